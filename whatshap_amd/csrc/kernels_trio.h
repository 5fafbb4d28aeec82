// kernels_trio.h -- trio run kernel (T = 4, three individuals): ped_tables, ped_cell / ped_column, resident_segment_ped.
// Included by dp_device.hip inside namespace whamd { namespace { ... } }: not a stand-alone header.
// ------------------------------------------------------------------------------------------------ trio runs
// Lookup tables of the trio runs: per column and individual two 64-entry tables (low / high 6 local cell bits) of
// L_s, computed once per solve at full-chip width.
__global__ __launch_bounds__(256) void ped_tables(const PedColumn* __restrict__ cols, uint32_t n_cols, int32_t* __restrict__ tables) {
	const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= n_cols * PED_TABLE) return;
	const uint32_t ci = idx / PED_TABLE, r = idx % PED_TABLE, s = r >> 7, half = (r >> 6) & 1u, v = r & 63u;
	const int32_t* __restrict__ d = cols[ci].dloc[s] + half * 6;
	int32_t sum = 0;
#pragma unroll
	for (int j = 0; j < 6; ++j) sum += ((v >> j) & 1u) ? d[j] : 0;
	tables[idx] = sum;
}

// value of lane k of the caller's quad (4 consecutive lanes), as a DPP move: no LDS traffic
template <int K>
__device__ __forceinline__ int32_t quad_bcast(int32_t v) {
	return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, false);
}

__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) { return __builtin_elementwise_add_sat(a, b); }

__device__ __forceinline__ int32_t sig_byte(uint32_t sig, int s) { return (int32_t)(sig << (24 - 8 * s)) >> 24; }

// One cell of a trio column for the calling lane (transmission value i = lane & 3): the cost of value i -- min over the
// lane's terms, NT of them in registers -- then the min-plus step over the previous value j
// (src/pedigreedptable.cpp:264-300).  The cost is the same for every j, so it is added after the argmin of
// P[j] + popcount(i ^ j) * recomb; INF stays INF through saturating adds.  Lm is the lane's own L_s, exchanged inside the quad.
template <int NT>
__device__ __forceinline__ uint32_t ped_cell(int32_t Lm, const uint4 p4, const uint32_t (&tc)[PED_REGTERMS],
                                             const int32_t (&s0)[PED_REGTERMS], const int32_t (&s1)[PED_REGTERMS],
                                             const int32_t (&s2)[PED_REGTERMS], const uint32_t (&rcj)[4],
                                             const uint2* pool, uint32_t tq0, uint32_t tq1, uint32_t& mj) {
	const int32_t L0 = quad_bcast<0>(Lm), L1 = quad_bcast<1>(Lm), L2 = quad_bcast<2>(Lm);
	uint32_t cost = 0xFFFFFFFFu;
#pragma unroll
	for (int k = 0; k < (NT < PED_REGTERMS ? NT : PED_REGTERMS); ++k)
		cost = min(cost, tc[k] + (uint32_t)(__mul24(s0[k], L0) + __mul24(s1[k], L1) + __mul24(s2[k], L2)));  // absent: c = INF, sig = 0
	if (NT > PED_REGTERMS) {  // more terms than registers (genotypes not trusted): the rest from the run's pool
		for (uint32_t q = tq0 + PED_REGTERMS; q < tq1; ++q) {
			const uint2 tm = pool[q];
			cost = min(cost, tm.x + (uint32_t)(__mul24(sig_byte(tm.y, 0), L0) + __mul24(sig_byte(tm.y, 1), L1) + __mul24(sig_byte(tm.y, 2), L2)));
		}
	}
	const uint32_t u0 = sat_add(p4.x, rcj[0]), u1 = sat_add(p4.y, rcj[1]), u2 = sat_add(p4.z, rcj[2]), u3 = sat_add(p4.w, rcj[3]);
	const uint32_t m01 = min(u0, u1), m23 = min(u2, u3);
	const uint32_t j01 = u1 < u0 ? 1u : 0u, j23 = u3 < u2 ? 3u : 2u;
	mj = m23 < m01 ? j23 : j01;
	return sat_add(min(m01, m23), cost);
}

// All cells of one trio column that project onto the calling lane's entries; NT as in ped_cell.
template <int NT>
__device__ __forceinline__ void ped_column(const uint32_t* lw, const int32_t* tbs, const uint2* pool, const uint4* bufP, uint4* bufQ,
                                           uint8_t* rec, uint32_t tid, uint32_t NTHR, uint32_t ti, uint32_t si,
                                           const uint32_t (&hop)[4]) {
	const uint4* hp = reinterpret_cast<const uint4*>(lw);
	// ---- LDS round 1: addresses depend on (column, lane) only
	const uint4 h0 = hp[0], h1 = hp[1], h2 = hp[2], h3 = hp[3], h4 = hp[4];
	const int32_t Sgm = (int32_t)lw[offsetof(PedColumn, Sg) / 4 + si];
	const int32_t dE0 = (int32_t)lw[offsetof(PedColumn, dE) / 4 + si], dE1 = (int32_t)lw[offsetof(PedColumn, dE) / 4 + 4 + si],
	              dE2 = (int32_t)lw[offsetof(PedColumn, dE) / 4 + 8 + si];
	const uint4* tp = reinterpret_cast<const uint4*>(lw + offsetof(PedColumn, rterms) / 4) + ti * 2;
	const uint4 ta = tp[0], tb4 = tp[1];
	uint32_t tq0 = 0, tq1 = 0;
	if (NT > PED_REGTERMS) { tq0 = h1.w + lw[offsetof(PedColumn, tptr) / 4 + ti]; tq1 = h1.w + lw[offsetof(PedColumn, tptr) / 4 + ti + 1]; }
	const uint32_t Lf = uni(h0.y), ebits = uni(h0.z);
	const uint32_t lowmask = h1.x, recomb = h1.y, PG = h4.x;
	const uint32_t epos0 = uni(h2.x), epos1 = uni(h2.y), epos2 = uni(h2.z);
	const uint32_t nlanes = 4u << Lf;
	const uint32_t tc[PED_REGTERMS] = {ta.x, ta.z, tb4.x, tb4.z};
	const uint32_t tsig[PED_REGTERMS] = {ta.y, ta.w, tb4.y, tb4.w};
	int32_t s0[PED_REGTERMS], s1[PED_REGTERMS], s2[PED_REGTERMS];
#pragma unroll
	for (int k = 0; k < PED_REGTERMS; ++k) { s0[k] = sig_byte(tsig[k], 0); s1[k] = sig_byte(tsig[k], 1); s2[k] = sig_byte(tsig[k], 2); }
	uint32_t rcj[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) rcj[j] = ((hop[j] & 1u) ? recomb : 0u) + ((hop[j] & 2u) ? 2u * recomb : 0u);
	for (uint32_t idx = tid; idx < nlanes; idx += NTHR) {
		uint32_t base = idx >> 2;
		if (ebits > 0) base = insert_zero(base, epos0);
		if (ebits > 1) base = insert_zero(base, epos1);
		if (ebits > 2) base = insert_zero(base, epos2);
		// ---- LDS round 2: tables and previous-slice rows
		const int32_t t_lo = tbs[base & 63u], t_hi = tbs[64 + ((base >> 6) & 63u)];
		uint32_t bD, bE = 0, bJ;
		if (ebits == 0) {
			const uint4 pa = bufP[base & lowmask];
			bD = ped_cell<NT>(Sgm + t_lo + t_hi, pa, tc, s0, s1, s2, rcj, pool, tq0, tq1, bJ);
			if (bD == 0xFFFFFFFFu) bJ = 0;
		} else if (ebits == 1) {
			const uint4 pa = bufP[base & lowmask], pb = bufP[(base | (1u << epos0)) & lowmask];
			const int32_t Lbase = Sgm + t_lo + t_hi;
			uint32_t j0, j1;
			const uint32_t A0 = ped_cell<NT>(Lbase, pa, tc, s0, s1, s2, rcj, pool, tq0, tq1, j0);
			const uint32_t A1 = ped_cell<NT>(Lbase + dE0, pb, tc, s0, s1, s2, rcj, pool, tq0, tq1, j1);
			// tie: the smaller Gray rank has x_h == parity of the bits above the ending read
			const uint32_t par = (PG ^ (uint32_t)__popc(base & h3.x)) & 1u;
			const bool take1 = A1 < sat_add(A0, par);
			bD = take1 ? A1 : A0;
			bE = take1 ? 1u : 0u;
			bJ = take1 ? j1 : j0;
			if (bD == 0xFFFFFFFFu) { bE = 0; bJ = 0; }
		} else {
			const int32_t Lbase = Sgm + t_lo + t_hi;
			const uint32_t ne = 1u << ebits;
			const uint32_t mLq[RES_EMAX] = {h3.x, h3.y, h3.z};
			bD = 0xFFFFFFFFu; bJ = 0;
			for (uint32_t e = 0; e < ne; ++e) {
				const uint32_t lc = base | ((e & 1u) << epos0) | (((e >> 1) & 1u) << epos1) | (((e >> 2) & 1u) << epos2);
				const int32_t Lm = Lbase + ((e & 1u) ? dE0 : 0) + ((e & 2u) ? dE1 : 0) + ((e & 4u) ? dE2 : 0);
				uint32_t mj;
				const uint32_t m = ped_cell<NT>(Lm, bufP[lc & lowmask], tc, s0, s1, s2, rcj, pool, tq0, tq1, mj);
				bool take = m < bD;
				if (e > 0 && m == bD && m != 0xFFFFFFFFu) {
					// the cells differ first (from the top) at ending read h; e ascends, so the new cell has x_h = 1
					const uint32_t h = 31u - (uint32_t)__clz((int)(e ^ bE));
					uint32_t par = 0;
#pragma unroll
					for (int q = 0; q < RES_EMAX; ++q) if (h == (uint32_t)q) par = ((PG >> q) ^ (uint32_t)__popc(lc & mLq[q])) & 1u;
					take = par != 0;
				}
				if (take) { bD = m; bE = e; bJ = mj; }
			}
		}
		reinterpret_cast<uint32_t*>(bufQ)[idx] = bD;
		rec[idx] = (uint8_t)(bE | (bJ << 3));
	}
}

// Resident run for a trio (T = 4 transmission values, three individuals; resident.h PedColumn).  Same run / grid-read /
// exchange machinery as resident_segment; a slice entry is the vector of T projection values.  Four consecutive lanes
// (a quad) own one projection entry, one lane per transmission value i.  Per cell projecting onto the entry:
//   * lane s < 3 looks up L_s (two 6-bit tables) and the quad exchanges the three sums with DPP moves;
//   * the lane's cost = min over ITS terms of c + sum_s sig_s L_s (24-bit multiply-adds); the first PED_REGTERMS terms
//     of every value come with the descriptor and stay in registers for the column, further ones (genotypes not
//     trusted) are read from the run's LDS pool;
//   * min-plus step and the argmin over the cells with the Gray-rank tie rule.
// Record: one byte per lane and column = ending-read bits | argj << 3.
// With two waves per SIMD (2^15 cells x 4 values = 2048 waves on 1024 SIMDs) nothing but the lane's own instruction
// stream hides LDS latency, so a column issues its LDS reads in two batches: everything addressed by (column, lane),
// then the table entries and previous-slice rows addressed by the cell index.
// SPEC: the run ends a backtrace chunk (ResSegment::in_mirror_bit carries the chunk's spec id for trio runs, which have no
// mirror) and leaves, per wave, the smallest entry (value << 32 | y * 4 + t) of the projection column it writes -- the seed of
// the speculative walk (kernels_backtrace.h).
template <bool DBG, bool SPEC = false>
__global__ __launch_bounds__(1024) void resident_segment_ped(DevProblem P, ResSegment sg, const uint32_t* __restrict__ prev,
                                                              uint32_t* __restrict__ cur) {
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	const unsigned long long t_begin = DBG ? __builtin_readcyclecounter() : 0ull;
	touch_kernel_arguments<sizeof(DevProblem) + sizeof(ResSegment) + 16>();
	const uint32_t w = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
	uint32_t* ldsc = smem;                                                           // ncols * PED_LDSWORDS
	int32_t* tab = reinterpret_cast<int32_t*>(ldsc + sg.ncols * PED_LDSWORDS);       // ncols * PED_TABLE words
	uint2* terms = reinterpret_cast<uint2*>(tab + sg.ncols * PED_TABLE);             // n_terms * 2 words
	uint4* bufP = reinterpret_cast<uint4*>(smem + ((sg.ncols * (PED_LDSWORDS + PED_TABLE) + sg.n_terms * 2 + 3) & ~3u));
	uint4* bufQ = bufP + (1u << sg.max_l);
	uint32_t* stage = reinterpret_cast<uint32_t*>(bufQ + (1u << sg.max_l));
	// per-column scalars that depend on the workgroup index, straight from the global descriptors: 64 lanes per column
	// = 16 lanes per individual (one per grid read) + 16 lanes for the tie-break parities of the grid part
	unsigned long long t_args = 0, t_first = 0;
	// per-column scalars that depend on the workgroup index, straight from the global descriptors: 64 lanes per column
	// = 16 lanes per individual (one per grid read) + 16 lanes for the tie-break parities of the grid part.  The raw
	// words are loaded here, in the same batch as everything else, and combined after the first barrier.
	uint32_t raw[2] = {0, 0};
	const uint32_t gs = (tid >> 4) & 3u, gq = tid & 15u;
	const bool graw = gs < (uint32_t)PED_NIND ? (gq < sg.g && ((w >> gq) & 1u)) : gq < (uint32_t)RES_EMAX;
#pragma unroll
	for (int u = 0; u < 2; ++u) {
		const uint32_t ci = u * (NT / 64) + (tid >> 6);
		if (ci < sg.ncols && graw) {
			const uint32_t* gw = reinterpret_cast<const uint32_t*>(P.ped_cols + sg.col_off + ci);
			raw[u] = gw[gs < (uint32_t)PED_NIND ? offsetof(PedColumn, dgrid) / 4 + gs * RES_GMAX + gq : offsetof(PedColumn, mG) / 4 + gq];
		}
	}
	{
		// descriptors (leading PED_LDSWORDS of each), tables, term pool, entering slice: batches of loads before the stores
		const uint4* __restrict__ gc = reinterpret_cast<const uint4*>(P.ped_cols + sg.col_off);
		const uint4* __restrict__ gt = reinterpret_cast<const uint4*>(P.ped_tables + (size_t)sg.col_off * PED_TABLE);
		const uint2* __restrict__ gq = reinterpret_cast<const uint2*>(P.ped_terms + sg.term_off);
		const uint4* __restrict__ p4 = reinterpret_cast<const uint4*>(prev);
		uint4* lc = reinterpret_cast<uint4*>(ldsc);
		uint4* lt = reinterpret_cast<uint4*>(tab);
		constexpr uint32_t DQ = PED_LDSWORDS / 4, GQ = sizeof(PedColumn) / 16;
		const uint32_t ndesc = sg.ncols * DQ, ntab = sg.ncols * (PED_TABLE / 4), nslice = sg.has_prev ? (1u << sg.Lb0) : 0u;
		const uint32_t wpart = deposit_args(w, sg.in_grid, sg.n_in_grid);
		if (DBG) { t_args = __builtin_readcyclecounter() + (wpart & 0u); }
		auto desc_at = [&](uint32_t i) { return gc[(i / DQ) * GQ + i % DQ]; };
		auto slice_at = [&](uint32_t l) { return p4[wpart | deposit_args(l, sg.in_local, sg.n_in_local)]; };
		// first batch: every load is issued before the first LDS store, so ONE memory latency covers the batch
		uint4 vd[2], vt[4], vs[2];
		uint2 vq[2];
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + tid; vd[u] = i < ndesc ? desc_at(i) : make_uint4(0, 0, 0, 0); }
#pragma unroll
		for (int u = 0; u < 4; ++u) { const uint32_t i = u * NT + tid; vt[u] = i < ntab ? gt[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + tid; vq[u] = i < sg.n_terms ? gq[i] : make_uint2(0, 0); }
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t l = u * NT + tid; vs[u] = l < nslice ? slice_at(l) : make_uint4(0, 0, 0, 0); }
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + tid; if (i < ndesc) lc[i] = vd[u]; }
		if (DBG) { t_first = __builtin_readcyclecounter(); }
#pragma unroll
		for (int u = 0; u < 4; ++u) { const uint32_t i = u * NT + tid; if (i < ntab) lt[i] = vt[u]; }
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + tid; if (i < sg.n_terms) terms[i] = vq[u]; }
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t l = u * NT + tid; if (l < nslice) bufP[l] = vs[u]; }
		// remainders (long runs of narrow workgroups, large term pools)
		for (uint32_t i = 2 * NT + tid; i < ndesc; i += NT) lc[i] = desc_at(i);
		for (uint32_t i = 4 * NT + tid; i < ntab; i += NT) lt[i] = gt[i];
		for (uint32_t i = 2 * NT + tid; i < sg.n_terms; i += NT) terms[i] = gq[i];
		for (uint32_t l = 2 * NT + tid; l < nslice; l += NT) bufP[l] = slice_at(l);
		if (!sg.has_prev && tid == 0) bufP[0] = make_uint4(0, 0, 0, 0);
	}
	const unsigned long long t_loaded = DBG ? __builtin_readcyclecounter() : 0ull;
	__syncthreads();
#pragma unroll
	for (int u = 0; u < 2; ++u) {
		const uint32_t ci = u * (NT / 64) + (tid >> 6), s = gs, q = gq;
		int32_t v = s < (uint32_t)PED_NIND ? (int32_t)raw[u] : (int32_t)((graw ? (uint32_t)__popc(w & raw[u]) & 1u : 0u) << q);
		v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
		if (ci < sg.ncols && q == 0) {
			PedColumn* pc = reinterpret_cast<PedColumn*>(ldsc + ci * PED_LDSWORDS);
			if (s < (uint32_t)PED_NIND) pc->Sg[s] = v; else pc->PG = (uint32_t)v;
		}
	}
	for (uint32_t ci0 = 2 * (NT / 64); ci0 < sg.ncols; ci0 += NT / 64) {  // long runs of narrow workgroups
		const uint32_t ci = ci0 + (tid >> 6), s = (tid >> 4) & 3u, q = tid & 15u;
		int32_t v = 0;
		if (ci < sg.ncols) {
			const PedColumn& gcol = P.ped_cols[sg.col_off + ci];
			if (s < (uint32_t)PED_NIND) { if (q < sg.g && ((w >> q) & 1u)) v = gcol.dgrid[s][q]; }
			else if (q < (uint32_t)RES_EMAX) v = (int32_t)(((uint32_t)__popc(w & gcol.mG[q]) & 1u) << q);
		}
		v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
		if (ci < sg.ncols && q == 0) {
			PedColumn* pc = reinterpret_cast<PedColumn*>(ldsc + ci * PED_LDSWORDS);
			if (s < (uint32_t)PED_NIND) pc->Sg[s] = v; else pc->PG = (uint32_t)v;
		}
	}
	__syncthreads();
	const unsigned long long t_ready = DBG ? __builtin_readcyclecounter() : 0ull;
	const uint32_t ti = tid & 3u;                       // this lane's transmission value
	const uint32_t si = ti < 3u ? ti : 2u;              // the individual whose L_s this lane looks up
	const uint32_t hop[4] = {(uint32_t)__popc(ti), (uint32_t)__popc(ti ^ 1u), (uint32_t)__popc(ti ^ 2u), (uint32_t)__popc(ti ^ 3u)};
	for (uint32_t ci = 0; ci < sg.ncols; ++ci) {
		const uint32_t* lw = ldsc + ci * PED_LDSWORDS;
		const uint32_t maxcnt = uni(lw[offsetof(PedColumn, maxcnt) / 4]);
		uint8_t* rec = reinterpret_cast<uint8_t*>(stage + uni(lw[offsetof(PedColumn, stage_off) / 4]));
		const int32_t* tbs = tab + ci * PED_TABLE + si * 128;
		if (maxcnt <= 2) ped_column<2>(lw, tbs, terms, bufP, bufQ, rec, tid, NT, ti, si, hop);
		else if (maxcnt <= (uint32_t)PED_REGTERMS) ped_column<PED_REGTERMS>(lw, tbs, terms, bufP, bufQ, rec, tid, NT, ti, si, hop);
		else ped_column<PED_REGTERMS + 1>(lw, tbs, terms, bufP, bufQ, rec, tid, NT, ti, si, hop);
		__syncthreads();
		uint4* tmp = bufP; bufP = bufQ; bufQ = tmp;
	}
	const unsigned long long t_cols = DBG ? __builtin_readcyclecounter() : 0ull;
	const uint32_t wout = deposit_args(w, sg.out_grid, sg.n_out_grid);
	uint4* c4 = reinterpret_cast<uint4*>(cur);
	unsigned long long best_key = ~0ull;
	for (uint32_t l = tid; l < (1u << sg.Lf_last); l += NT) {
		const uint32_t y = wout | deposit_args(l, sg.out_local, sg.n_out_local);
		const uint4 v = bufP[l];
		c4[y] = v;
		if (SPEC) {
			best_key = min(best_key, ((unsigned long long)v.x << 32) | (y * 4u + 0u));
			best_key = min(best_key, ((unsigned long long)v.y << 32) | (y * 4u + 1u));
			best_key = min(best_key, ((unsigned long long)v.z << 32) | (y * 4u + 2u));
			best_key = min(best_key, ((unsigned long long)v.w << 32) | (y * 4u + 3u));
		}
	}
	if (SPEC && sg.in_mirror_bit) {
#pragma unroll
		for (int m = 1; m < 64; m <<= 1) {
			const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best_key, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best_key >> 32), m);
			best_key = min(best_key, ((unsigned long long)hi << 32) | lo);
		}
		if ((tid & 63u) == 0) P.spec_keys[(size_t)(sg.in_mirror_bit - 1u) * P.spec_stride + w * (NT >> 6) + (tid >> 6)] = best_key;
	}
	unsigned long long* grec = reinterpret_cast<unsigned long long*>(P.bt + (((unsigned long long)sg.bt_hi << 32) | sg.bt_lo)) + (size_t)w * sg.stage_words;
	const unsigned long long* st64 = reinterpret_cast<const unsigned long long*>(stage);
	for (uint32_t i = tid; i < sg.stage_words; i += NT) grec[i] = st64[i];
	if (DBG && w == 0 && tid == 0) {
		unsigned long long* d = P.dbg + (size_t)sg.pad * 8;
		d[0] = t_ready - t_begin;
		d[1] = t_cols - t_ready;
		d[2] = __builtin_readcyclecounter() - t_cols;
		d[3] = sg.ncols;
		d[4] = (t_loaded - t_begin) * sg.ncols; d[5] = (t_args - t_begin) * sg.ncols; d[6] = (t_first - t_begin) * sg.ncols;
		d[7] = sg.ncols;
	}
}
