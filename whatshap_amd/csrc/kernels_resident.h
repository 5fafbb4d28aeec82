// kernels_resident.h -- single-individual run kernel (resident.h): resident_tables, the 32-bit and the packed 16-bit column evaluators, resident_segment.
// Included by dp_device.hip inside namespace whamd { namespace { ... } }: not a stand-alone header.
// ------------------------------------------------------------------------------------------------ resident run
// One launch = one run of consecutive columns (resident.h).  Workgroup w owns the slice of the projection column whose
// grid-read bits equal w; the slice lives in LDS (two buffers), Pr touches HBM only at the load and the store.
// Single individual (T = 1): cost(x) = min(Cp + S, Cm - S, Cc), S = S_grid(w) + tab_lo[l & 127] + tab_hi[l >> 7].
// Everything a column needs (descriptor, lookup tables) is staged in LDS before the first column, so the sequential
// column chain contains no global-memory latency.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// The by-value arguments of the run kernels span several 64-byte lines and the compiler fetches them with one scalar
// load per use, waiting each time: ~3600 cycles (1.5 us) of serialized scalar-cache misses at the start of every run
// (measured, scripts/gpu_timing_trio.py).  Touching every line with independent loads first costs one miss latency.
template <int BYTES>
__device__ __forceinline__ void touch_kernel_arguments() {
	typedef const __attribute__((address_space(4))) uint32_t* karg_ptr;
	const karg_ptr ka = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
	uint32_t acc = 0;
#pragma unroll
	for (int l = 0; l < (BYTES + 63) / 64; ++l) acc |= ka[l * 16];
	asm volatile("" ::"s"(acc));
}

// `segs` is one of the RES_IOSEG-word run arrays of the kernel arguments.  Fully unrolled with static indices: the
// words are fetched with one wide scalar load and stay in SGPRs; a loop with a dynamic trip count made the compiler
// fetch every word with its own scalar load and wait for it, at every use (~18 serialized loads per run prologue).
__device__ __forceinline__ uint32_t deposit_args(uint32_t v, const uint32_t (&segs)[RES_IOSEG], uint32_t nseg) {
	uint32_t x = 0;
#pragma unroll
	for (uint32_t i = 0; i < (uint32_t)RES_IOSEG; ++i) {
		const uint32_t sg = segs[i];
		const uint32_t piece = ((v >> (sg & 31u)) & ((1u << ((sg >> 16) & 31u)) - 1u)) << ((sg >> 8) & 31u);
		x |= i < nseg ? piece : 0u;
	}
	return x;
}

constexpr int RES_OPT = 2;  // generic path: projection entries a thread evaluates together

// local cell index with a zero inserted at bit position p
__device__ __forceinline__ uint32_t insert_zero(uint32_t v, uint32_t p) {
	return ((v >> p) << (p + 1u)) | (v & ((1u << p) - 1u));
}

// min(Cp + S, Cm - S, Cc): an absent plus/minus term is RES_ABSENT and can never be the minimum (resident.h)
__device__ __forceinline__ uint32_t res_cost(uint32_t Cp, uint32_t Cm, uint32_t Cc, int32_t S) {
	return min(min(Cp + (uint32_t)S, Cm - (uint32_t)S), Cc);
}

// Lookup tables of the local part of S for every resident column (two 128-entry tables: low / high 7 local bits),
// computed once per solve at full-chip width; a run copies its columns' tables into LDS.
__global__ __launch_bounds__(256) void resident_tables(const ResColumn* __restrict__ cols, uint32_t n_cols, int32_t* __restrict__ tables) {
	const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= n_cols * RES_TABLE) return;
	const uint32_t ci = idx >> 8, half = (idx >> 7) & 1u, v = idx & 127u;
	const int32_t* __restrict__ d = cols[ci].dloc + half * 7;
	int32_t sum = 0;
#pragma unroll
	for (int j = 0; j < 7; ++j) sum += ((v >> j) & 1u) ? d[j] : 0;
	tables[idx] = sum;
}

// One vectorised column of a resident run for the calling thread's entries (resident.h RES_MODE_E0 .. E1_BIT1), with
// the costs of up to RES_MAXFOLD preceding folded columns added per cell.  A thread owns the 4 consecutive entries
// 4t .. 4t+3 (8 cells when a read ends) and moves them with 16-byte LDS accesses.
// All LDS reads of a step (slice entries, the records and table lookups of this column and of the first folded
// column) are issued before the first use, so one LDS latency covers them instead of one per folded column.
//
// Shared tail of both variants: per-entry minimum over the (up to two) cells with the Gray-rank tie rule, slice store,
// one record byte per thread (bit u = argmin side of the ending read for entry 4t+u).
template <uint32_t MODE, int NC, bool MIRROR>
__device__ __forceinline__ void res_finish_entries(const uint32_t (&acc)[NC], uint32_t base, uint32_t mL0, uint32_t PG, uint32_t pbits,
                                                   uint32_t* bufQ, uint8_t* rec, uint32_t t) {
	uint32_t D[4];
	uint32_t takes = 0;
	if (MODE == RES_MODE_E0) {
#pragma unroll
		for (int u = 0; u < 4; ++u) D[u] = acc[u];
	} else {
		// tie: the smaller Gray rank has x_h == parity of the bits above the ending read; bit u of parx is
		// that parity for entry 4t+u (grid part PG, this thread's part, the per-entry constant pbits)
		const uint32_t parx = (0u - ((PG ^ (uint32_t)__popc(base & mL0)) & 1u)) ^ pbits;
		const uint32_t mflip = (pbits >> 8) & 1u;
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			// cell pair of entry 4t+u: E1_HIGH (u, 4+u); E1_BIT0 (2u, 2u+1); E1_BIT1 ((u>>1)*4 + (u&1), +2)
			const int c0 = MODE == RES_MODE_E1_HIGH ? u : (MODE == RES_MODE_E1_BIT0 ? 2 * u : (((u >> 1) << 2) | (u & 1)));
			const int c1i = MODE == RES_MODE_E1_HIGH ? 4 + u : (MODE == RES_MODE_E1_BIT0 ? 2 * u + 1 : c0 + 2);
			const uint32_t par = (parx >> u) & 1u;
			const uint32_t A0 = acc[c0 & (NC - 1)], A1 = acc[c1i & (NC - 1)];
			// side 1 wins if strictly smaller, or equal and favoured by the tie rule: A1 < A0 + par
			D[u] = min(A0, A1);
			takes |= (A1 < A0 + par) ? (1u << u) : 0u;
			// the mirror image of this entry (all bits complemented) has the two sides swapped and its own parity:
			// its decision goes to bit 4 + u (read by the backtrace when the path runs through the half not computed)
			if (MIRROR) takes |= (A0 < A1 + (par ^ mflip)) ? (16u << u) : 0u;
		}
	}
	*reinterpret_cast<uint4*>(bufQ + (t << 2)) = make_uint4(D[0], D[1], D[2], D[3]);
	if (MODE != RES_MODE_E0) rec[t] = (uint8_t)takes;
}

// 32-bit evaluation (columns without pk_ok): reads the words 16..35 of the descriptors.
template <uint32_t MODE>
__device__ __forceinline__ void res_fast_column(const uint32_t* ldsc, const int32_t* tab, uint32_t ci, uint32_t nfold,
                                                const uint32_t* bufP, uint32_t* bufQ, uint8_t* stage, uint32_t tid,
                                                uint32_t NT, uint32_t nthr, const uint4 q2) {
	constexpr int NC = MODE == RES_MODE_E0 ? 4 : 8;  // cells per thread
	constexpr int H0 = offsetof(ResColumn, Cp) / 16;
	const uint4* hp = reinterpret_cast<const uint4*>(ldsc + ci * RES_LDSWORDS);
	const uint4 h0 = hp[H0], h2 = hp[H0 + 1], h3 = hp[H0 + 2], h4 = hp[H0 + 3], h5 = hp[H0 + 4];
	const uint32_t lowmask = q2.x, ep0 = h2.x, mL0 = h3.x, PG = h4.y;
	const uint32_t pbits = ldsc[ci * RES_LDSWORDS + offsetof(ResColumn, pbits) / 4];
	uint8_t* rec = stage + q2.z * 8u;
	// record of the first folded column (or of this column again when nothing is folded: loaded but not used)
	const uint32_t c1 = ci - (nfold ? 1u : 0u);
	const uint4* gp = reinterpret_cast<const uint4*>(ldsc + c1 * RES_LDSWORDS);
	const uint4 g0 = gp[H0], g4 = gp[H0 + 3], g5 = gp[H0 + 4];
	const int32_t* tl0 = tab + ci * RES_TABLE;
	const int32_t* tl1 = tab + c1 * RES_TABLE;
	const int32_t* dl0 = reinterpret_cast<const int32_t*>(ldsc + ci * RES_LDSWORDS + offsetof(ResColumn, dloc) / 4);
	const int32_t* dl1 = reinterpret_cast<const int32_t*>(ldsc + c1 * RES_LDSWORDS + offsetof(ResColumn, dloc) / 4);
	for (uint32_t t = tid; t < nthr; t += NT) {
		const uint32_t l4 = t << 2;
		uint32_t base, base1 = 0;
		if (MODE == RES_MODE_E0) base = l4;
		else if (MODE == RES_MODE_E1_HIGH) { base = insert_zero(l4, ep0); base1 = base | (1u << ep0); }
		else base = l4 << 1;
		// ---- issue every LDS read of this thread
		uint4 pa, pb = make_uint4(0, 0, 0, 0);
		pa = *reinterpret_cast<const uint4*>(bufP + (base & lowmask));
		if (MODE == RES_MODE_E1_HIGH) pb = *reinterpret_cast<const uint4*>(bufP + (base1 & lowmask));
		else if (MODE != RES_MODE_E0) pb = *reinterpret_cast<const uint4*>(bufP + ((base + 4u) & lowmask));
		const uint32_t ilo = base & 127u, ihi = 128u + ((base >> 7) & 127u);
		const int32_t ta0 = tl0[ilo], tb0 = tl0[ihi], ta1 = tl1[ilo], tb1 = tl1[ihi];
		int32_t dE0 = 0, dE1 = 0;
		if (MODE == RES_MODE_E1_HIGH) { dE0 = dl0[ep0]; dE1 = dl1[ep0]; }  // delta of the ending read (0 where it was not active yet)
		uint32_t acc[NC];
		acc[0] = pa.x; acc[1] = pa.y; acc[2] = pa.z; acc[3] = pa.w;
		if constexpr (NC == 8) { acc[4] = pb.x; acc[5] = pb.y; acc[6] = pb.z; acc[7] = pb.w; }
		// ---- acc[c] += cost_column(cell c) for this column and the folded ones before it
		// cost(S) = min3(Cp + S, Cm - S, Cc) with A = Cp + S: min3(A, (Cp + Cm) - A, Cc) -- one add per cell after the base
		auto add_column = [&](const uint4 f0, const uint4 f4, const uint4 f5, int32_t ta, int32_t tb, int32_t dE) {
			const uint32_t K = f0.x + f0.y, Cc = f0.z;
			const uint32_t A0 = f0.x + (uint32_t)((int32_t)f4.x + ta + tb);
			const uint32_t d0 = f5.x, d1 = f5.y, d2 = f5.z;
			auto cell = [&](uint32_t A) -> uint32_t { return min(min(A, K - A), Cc); };
			if (MODE == RES_MODE_E0) {
				const uint32_t pat[4] = {0, d0, d1, d0 + d1};
#pragma unroll
				for (int c = 0; c < 4; ++c) acc[c] += cell(A0 + pat[c]);
			} else if (MODE == RES_MODE_E1_HIGH) {
				const uint32_t pat[4] = {0, d0, d1, d0 + d1};
				const uint32_t A1 = A0 + (uint32_t)dE;
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					acc[c] += cell(A0 + pat[c]);
					if constexpr (NC == 8) acc[4 + c] += cell(A1 + pat[c]);
				}
			} else {
				const uint32_t pat[8] = {0, d0, d1, d0 + d1, d2, d2 + d0, d2 + d1, d2 + d0 + d1};
#pragma unroll
				for (int c = 0; c < NC; ++c) acc[c] += cell(A0 + pat[c & 7]);
			}
		};
		add_column(h0, h4, h5, ta0, tb0, dE0);
		if (nfold) add_column(g0, g4, g5, ta1, tb1, dE1);
		for (uint32_t f = 2; f <= nfold; ++f) {  // further folded columns (rare)
			const uint32_t cf = ci - f;
			const uint4* fp = reinterpret_cast<const uint4*>(ldsc + cf * RES_LDSWORDS);
			const int32_t* tlf = tab + cf * RES_TABLE;
			int32_t dEf = 0;
			if (MODE == RES_MODE_E1_HIGH) dEf = reinterpret_cast<const int32_t*>(ldsc + cf * RES_LDSWORDS + offsetof(ResColumn, dloc) / 4)[ep0];
			add_column(fp[H0], fp[H0 + 3], fp[H0 + 4], tlf[ilo], tlf[ihi], dEf);
		}
		res_finish_entries<MODE, NC, true>(acc, base, mL0, PG, pbits, bufQ, rec, t);  // rare path: always both decisions
	}
}

// Packed 16-bit evaluation (resident.h pk_ok): two cells per instruction, the costs of the folded columns are added in
// 16 bits and widened once.  Reads only Q0..Q3 of this column and Q0, Q1 of every folded one.
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_pk(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// MIRROR: the run is halved (ResSegment::half) -- records carry the mirror-image decisions as well.
template <uint32_t MODE, bool MIRROR>
__device__ __forceinline__ void res_pk_column(const uint32_t* ldsc, const int32_t* tab, uint32_t ci, uint32_t nfold,
                                              const uint32_t* bufP, uint32_t* bufQ, uint8_t* stage, uint32_t tid,
                                              uint32_t NT, uint32_t nthr, const uint4 q0, const uint4 q2) {
	constexpr int NC = MODE == RES_MODE_E0 ? 4 : 8;  // cells per thread
	const uint4* hp = reinterpret_cast<const uint4*>(ldsc + ci * RES_LDSWORDS);
	const uint4 q1 = hp[1], q3 = hp[3];
	const uint32_t lowmask = q2.x, ep0 = q3.x, mL0 = q3.y, PG = q3.z;
	uint8_t* rec = stage + q2.z * 8u;
	// the first folded column is evaluated unconditionally (with nothing folded it is this column again) and masked:
	// its reads share the LDS round of the others and the loop body has no branch
	const uint32_t c1 = ci - (nfold ? 1u : 0u);
	const uint4* gp = reinterpret_cast<const uint4*>(ldsc + c1 * RES_LDSWORDS);
	const uint4 g0 = gp[0], g1 = gp[1];
	const u16x2 keep1 = as_pk(nfold ? 0xFFFFFFFFu : 0u);
	const int32_t* tl0 = tab + ci * RES_TABLE;
	const int32_t* tl1 = tab + c1 * RES_TABLE;
	const int32_t* dl0 = reinterpret_cast<const int32_t*>(ldsc + ci * RES_LDSWORDS + offsetof(ResColumn, dloc) / 4);
	const int32_t* dl1 = reinterpret_cast<const int32_t*>(ldsc + c1 * RES_LDSWORDS + offsetof(ResColumn, dloc) / 4);
	for (uint32_t t = tid; t < nthr; t += NT) {
		const uint32_t l4 = t << 2;
		uint32_t base, base1 = 0;
		if (MODE == RES_MODE_E0) base = l4;
		else if (MODE == RES_MODE_E1_HIGH) { base = insert_zero(l4, ep0); base1 = base | (1u << ep0); }
		else base = l4 << 1;
		uint4 pa, pb = make_uint4(0, 0, 0, 0);
		pa = *reinterpret_cast<const uint4*>(bufP + (base & lowmask));
		if (MODE == RES_MODE_E1_HIGH) pb = *reinterpret_cast<const uint4*>(bufP + (base1 & lowmask));
		else if (MODE != RES_MODE_E0) pb = *reinterpret_cast<const uint4*>(bufP + ((base + 4u) & lowmask));
		const uint32_t ilo = base & 127u, ihi = 128u + ((base >> 7) & 127u);
		const int32_t ta0 = tl0[ilo], tb0 = tl0[ihi], ta1 = tl1[ilo], tb1 = tl1[ihi];
		int32_t dE0 = 0, dE1 = 0;
		if (MODE == RES_MODE_E1_HIGH) { dE0 = dl0[ep0]; dE1 = dl1[ep0]; }
		uint32_t acc[NC];
		acc[0] = pa.x; acc[1] = pa.y; acc[2] = pa.z; acc[3] = pa.w;
		if constexpr (NC == 8) { acc[4] = pb.x; acc[5] = pb.y; acc[6] = pb.z; acc[7] = pb.w; }
		u16x2 tot[NC / 2];
		auto pk_column = [&](const uint4 f0, const uint4 f1, int32_t ta, int32_t tb, int32_t dE, bool first, u16x2 keep) {
			const uint32_t A0 = f0.x + (uint32_t)(ta + tb);  // Cp + S_grid + local part, < 2^14
			const u16x2 K = as_pk(f0.y), Cc = as_pk(f0.z);
			const u16x2 A0p = as_pk((A0 & 0xFFFFu) | (A0 << 16));
			auto cell2 = [&](u16x2 A) -> u16x2 { return __builtin_elementwise_min(__builtin_elementwise_min(A, K - A), Cc); };
			u16x2 cst[NC / 2];
			cst[0] = cell2(A0p + as_pk(f1.x)); cst[1] = cell2(A0p + as_pk(f1.y));
			if constexpr (NC == 8) {
				if (MODE == RES_MODE_E1_HIGH) {
					const uint32_t A1 = A0 + (uint32_t)dE;
					const u16x2 A1p = as_pk((A1 & 0xFFFFu) | (A1 << 16));
					cst[2] = cell2(A1p + as_pk(f1.x)); cst[3] = cell2(A1p + as_pk(f1.y));
				} else {
					cst[2] = cell2(A0p + as_pk(f1.z)); cst[3] = cell2(A0p + as_pk(f1.w));
				}
			}
#pragma unroll
			for (int i = 0; i < NC / 2; ++i) tot[i] = first ? cst[i] : tot[i] + (cst[i] & keep);
		};
		const u16x2 all = as_pk(0xFFFFFFFFu);
		pk_column(q0, q1, ta0, tb0, dE0, true, all);
		pk_column(g0, g1, ta1, tb1, dE1, false, keep1);
		for (uint32_t f = 2; f <= nfold; ++f) {  // further folded columns (rare)
			const uint32_t cf = ci - f;
			const uint4* fp = reinterpret_cast<const uint4*>(ldsc + cf * RES_LDSWORDS);
			const int32_t* tlf = tab + cf * RES_TABLE;
			int32_t dEf = 0;
			if (MODE == RES_MODE_E1_HIGH) dEf = reinterpret_cast<const int32_t*>(ldsc + cf * RES_LDSWORDS + offsetof(ResColumn, dloc) / 4)[ep0];
			pk_column(fp[0], fp[1], tlf[ilo], tlf[ihi], dEf, false, all);
		}
#pragma unroll
		for (int i = 0; i < NC / 2; ++i) { acc[2 * i] += (uint32_t)tot[i].x; acc[2 * i + 1] += (uint32_t)tot[i].y; }
		res_finish_entries<MODE, NC, MIRROR>(acc, base, mL0, PG, q3.w, bufQ, rec, t);
	}
}

// PMC finding (profiles/r01_pmc_resident_v1.txt): the per-column loop is bound by instruction ISSUE, first of all by the
// scalar unit the 16 waves of a workgroup share -- so the loop keeps per-column constants in vector registers (LDS
// broadcast reads), lets whole waves without work branch straight to the barrier, and records the argmin bits as one
// byte per thread (no ballot / exec-mask sequences).
// Body of one run for workgroup `w` (shared by the one-run launch and the batched launch below).  `score_out` != nullptr:
// the run ends a connected component -- workgroup 0 stores the single exit value there (DeviceTable jobs).
// SYM: the run takes part in the complement symmetry (it is halved, reads what a halved run wrote, or must store the
// mirror image); runs that do not are launched without that code (it costs ~4 % where a run is latency-bound).
template <bool DBG, bool SYM>
__device__ __forceinline__ void resident_segment_body(const DevProblem& P, const ResSegment& sg, const uint32_t* __restrict__ prev,
                                                      uint32_t* __restrict__ cur, const uint32_t w, uint32_t* score_out,
                                                      const unsigned long long t_begin) {
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	const uint32_t tid = threadIdx.x, NT = blockDim.x;
	const unsigned long long rt_begin = DBG ? wall_clock64() : 0ull;
	unsigned long long t_args = 0, t_first = 0;
	uint32_t* ldsc = smem;                                             // ncols * 64 words: column descriptors
	int32_t* tab = reinterpret_cast<int32_t*>(smem + sg.ncols * 64);   // ncols * 256 words: lookup tables
	uint32_t* bufP = smem + sg.ncols * (64 + RES_TABLE);
	uint32_t* bufQ = bufP + (1u << sg.max_l);
	uint8_t* stage = reinterpret_cast<uint8_t*>(bufQ + (1u << sg.max_l));  // backtrace record of the run (stage_words * 8 bytes)
	// per-column scalars that depend on the workgroup index, straight from the global descriptors: 16 lanes per column
	// (one per grid read; lanes 0..2 also one tie-break mask each).  The raw words are loaded in the same batch as
	// everything else and combined after the first barrier.
	constexpr uint32_t GQ = sizeof(ResColumn) / 16, DQ = RES_LDSWORDS / 4;
	const uint32_t gi = tid & 15u;
	int32_t rawd[2] = {0, 0};
	uint32_t rawm[2] = {0, 0};
#pragma unroll
	for (int u = 0; u < 2; ++u) {
		const uint32_t ci = u * (NT / 16) + (tid >> 4);
		if (ci < sg.ncols) {
			const ResColumn& gcol = P.res_cols[sg.col_off + ci];
			if (gi < sg.g && ((w >> gi) & 1u)) rawd[u] = gcol.dgrid[gi];
			if (gi < (uint32_t)RES_EMAX) rawm[u] = gcol.mG[gi];
		}
	}
	// stage descriptors + lookup tables (coalesced 16-byte copies) and the entering slice (from the exchange layout).
	// All global loads of a batch are issued before the first LDS store, so one memory latency covers the batch.
	{
		const uint4* __restrict__ gc = reinterpret_cast<const uint4*>(P.res_cols + sg.col_off);
		const uint4* __restrict__ gt = reinterpret_cast<const uint4*>(P.res_tables + (size_t)sg.col_off * RES_TABLE);
		uint4* lc = reinterpret_cast<uint4*>(ldsc);
		uint4* lt = reinterpret_cast<uint4*>(tab);
		const uint32_t ndesc = sg.ncols * DQ, ntab = sg.ncols * (RES_TABLE / 4), nslice = sg.has_prev ? (1u << sg.Lb0) : 0u;
		const uint32_t wpart = deposit_args(w, sg.in_grid, sg.n_in_grid);
		if (DBG) t_args = __builtin_readcyclecounter() + (wpart & 0u);
		auto desc_at = [&](uint32_t i) { return gc[(i / DQ) * GQ + i % DQ]; };  // the leading RES_LDSWORDS of every descriptor
		// entering slice element l; after a halved run only the entries with a clear mirror bit exist: the others are read
		// from their complement (D[~x] == D[x])
		const uint32_t in_flip = (SYM && sg.in_half) ? sg.in_fullmask : 0u, in_bit = sg.in_mirror_bit & 31u;
		auto in_index = [&](uint32_t l) {
			const uint32_t idx = wpart | deposit_args(l, sg.in_local, sg.n_in_local);
			return (SYM && ((idx >> in_bit) & 1u)) ? idx ^ in_flip : idx;
		};
		uint4 vd[2], vt[4];
		uint32_t vs[4];
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + tid; vd[u] = i < ndesc ? desc_at(i) : make_uint4(0, 0, 0, 0); }
#pragma unroll
		for (int u = 0; u < 4; ++u) { const uint32_t i = u * NT + tid; vt[u] = i < ntab ? gt[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t l = u * NT + tid;
			vs[u] = l < nslice ? prev[in_index(l)] : 0u;
		}
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + tid; if (i < ndesc) lc[i] = vd[u]; }
		if (DBG) t_first = __builtin_readcyclecounter();
#pragma unroll
		for (int u = 0; u < 4; ++u) { const uint32_t i = u * NT + tid; if (i < ntab) lt[i] = vt[u]; }
#pragma unroll
		for (int u = 0; u < 4; ++u) { const uint32_t l = u * NT + tid; if (l < nslice) bufP[l] = vs[u]; }
		// remainders (long runs with few threads)
		for (uint32_t i = 2 * NT + tid; i < ndesc; i += NT) lc[i] = desc_at(i);
		for (uint32_t i = 4 * NT + tid; i < ntab; i += NT) lt[i] = gt[i];
		for (uint32_t l = 4 * NT + tid; l < nslice; l += NT) bufP[l] = prev[in_index(l)];
		if (!sg.has_prev && tid == 0) bufP[0] = 0;
	}
	const unsigned long long t_loaded = DBG ? __builtin_readcyclecounter() : 0ull;
	__syncthreads();
	// xor-shuffle reduce of the workgroup-dependent scalars; lane 0 of each 16 patches the staged descriptor
	auto patch_column = [&](uint32_t ci, int32_t part, uint32_t mraw) {
		uint32_t pg = gi < (uint32_t)RES_EMAX ? (((uint32_t)__popc(w & mraw) & 1u) << gi) : 0u;
		part += __shfl_xor(part, 1); part += __shfl_xor(part, 2); part += __shfl_xor(part, 4); part += __shfl_xor(part, 8);
		pg |= __shfl_xor(pg, 1); pg |= __shfl_xor(pg, 2); pg |= __shfl_xor(pg, 4); pg |= __shfl_xor(pg, 8);
		if (ci < sg.ncols && gi == 0) {
			ResColumn* rc = reinterpret_cast<ResColumn*>(ldsc + ci * RES_LDSWORDS);
			rc->Sg = part;
			rc->PG = pg;
			rc->A += (uint32_t)part;
			rc->PGq = pg;
		}
	};
#pragma unroll
	for (int u = 0; u < 2; ++u) patch_column(u * (NT / 16) + (tid >> 4), rawd[u], rawm[u]);
	for (uint32_t ci0 = 2 * (NT / 16); ci0 < sg.ncols; ci0 += NT / 16) {  // long runs of narrow workgroups
		const uint32_t ci = ci0 + (tid >> 4);
		int32_t part = 0;
		uint32_t mraw = 0;
		if (ci < sg.ncols) {
			const ResColumn& gcol = P.res_cols[sg.col_off + ci];
			if (gi < sg.g && ((w >> gi) & 1u)) part = gcol.dgrid[gi];
			if (gi < (uint32_t)RES_EMAX) mraw = gcol.mG[gi];
		}
		patch_column(ci, part, mraw);
	}
	__syncthreads();
	const unsigned long long t_ready = DBG ? __builtin_readcyclecounter() : 0ull;
	const uint32_t wave_first = tid & ~63u;  // first thread index of this wave
	unsigned long long acc_a = 0, acc_b = 0, acc_c = 0, nsteps_dbg = 0;
	for (uint32_t ci = 0; ci < sg.ncols; ++ci) {
		const unsigned long long tq0 = DBG ? __builtin_readcyclecounter() : 0ull;
		// hot words as LDS broadcasts into VECTOR registers (resident.h); only the flags / nthr become scalars
		const uint4* hp = reinterpret_cast<const uint4*>(ldsc + ci * RES_LDSWORDS);
		const uint4 q0 = hp[0], q2 = hp[2];
		const uint32_t flags = uni(q0.w), nthr = uni(q2.y);
		const uint32_t mode = flags & 255u;
		if (mode == RES_MODE_FOLDED) continue;  // evaluated inside the next vectorised column: no slice traffic, no barrier
		const unsigned long long tq1 = DBG ? __builtin_readcyclecounter() : 0ull;
		if (mode != RES_MODE_GENERIC) {
			if (wave_first < nthr) {  // a wave whose 64 threads all lie beyond nthr goes straight to the barrier
				const uint32_t nfold = (flags >> 8) & 15u;
				if (flags & (1u << 12)) {
					if (mode == RES_MODE_E0) res_pk_column<RES_MODE_E0, false>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
					else if (SYM && sg.half) {
						if (mode == RES_MODE_E1_HIGH) res_pk_column<RES_MODE_E1_HIGH, true>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
						else if (mode == RES_MODE_E1_BIT0) res_pk_column<RES_MODE_E1_BIT0, true>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
						else res_pk_column<RES_MODE_E1_BIT1, true>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
					} else {
						if (mode == RES_MODE_E1_HIGH) res_pk_column<RES_MODE_E1_HIGH, false>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
						else if (mode == RES_MODE_E1_BIT0) res_pk_column<RES_MODE_E1_BIT0, false>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
						else res_pk_column<RES_MODE_E1_BIT1, false>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
					}
				} else {
					if (mode == RES_MODE_E0) res_fast_column<RES_MODE_E0>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q2);
					else if (mode == RES_MODE_E1_HIGH) res_fast_column<RES_MODE_E1_HIGH>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q2);
					else if (mode == RES_MODE_E1_BIT0) res_fast_column<RES_MODE_E1_BIT0>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q2);
					else res_fast_column<RES_MODE_E1_BIT1>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q2);
				}
			}
		} else {
			constexpr int H0 = offsetof(ResColumn, Cp) / 16;
			const uint4 h0 = hp[H0], h1 = q2;
			const int32_t* tlo = tab + ci * RES_TABLE;
			const int32_t* thi = tlo + 128;
			const uint4 h2 = hp[H0 + 1], h3 = hp[H0 + 2], h4 = hp[H0 + 3];
			const uint32_t Cp = h0.x, Cm = h0.y, Cc = h0.z, lowmask = h1.x;
			const int32_t Sg = (int32_t)h4.x;
			const uint32_t PG = h4.y;
			const uint32_t Lf = uni(h4.w), ebits = uni(ldsc[ci * RES_LDSWORDS + offsetof(ResColumn, ebits) / 4]);
			const uint32_t nout = 1u << Lf;
			const uint32_t epos[RES_EMAX] = {uni(h2.x), uni(h2.y), uni(h2.z)};
			const uint32_t mL[RES_EMAX] = {uni(h3.x), uni(h3.y), uni(h3.z)};
			const uint32_t nw = uni(h1.w);
			unsigned long long* planes = reinterpret_cast<unsigned long long*>(stage) + uni(h1.z);
			for (uint32_t l0 = 0; l0 < nout; l0 += NT * RES_OPT) {
				uint32_t l_out[RES_OPT], base[RES_OPT], bestD[RES_OPT], beste[RES_OPT];
				bool valid[RES_OPT];
				uint32_t ebit[RES_EMAX];
#pragma unroll
				for (int q = 0; q < RES_EMAX; ++q) ebit[q] = (uint32_t)q < ebits ? (1u << epos[q]) : 0u;
#pragma unroll
				for (int u = 0; u < RES_OPT; ++u) {
					l_out[u] = l0 + u * NT + tid;
					valid[u] = l_out[u] < nout;
					base[u] = valid[u] ? l_out[u] : 0u;
					beste[u] = 0;
#pragma unroll
					for (int q = 0; q < RES_EMAX; ++q) if ((uint32_t)q < ebits) base[u] = insert_zero(base[u], epos[q]);
					bestD[u] = 0xFFFFFFFFu;
				}
				const uint32_t ne = 1u << ebits;
#pragma unroll
				for (uint32_t e = 0; e < (1u << RES_EMAX); ++e) {
					if (e < ne) {
#pragma unroll
						for (int u = 0; u < RES_OPT; ++u) {
							uint32_t lc = base[u];
#pragma unroll
							for (int q = 0; q < RES_EMAX; ++q) lc |= ((e >> q) & 1u) ? ebit[q] : 0u;
							const int32_t S = Sg + tlo[lc & 127u] + thi[(lc >> 7) & 127u];
							const uint32_t D = res_cost(Cp, Cm, Cc, S) + bufP[lc & lowmask];
							bool take = D < bestD[u];
							if (e > 0 && D == bestD[u]) {
								// candidates differ first (from the top) at ending read h; e ascends, so the new one has x_h = 1
								const uint32_t h = 31u - (uint32_t)__clz((int)(e ^ beste[u]));
								uint32_t par = 0;
#pragma unroll
								for (int q = 0; q < RES_EMAX; ++q)
									if (h == (uint32_t)q) par = ((PG >> q) ^ (uint32_t)__popc(lc & mL[q])) & 1u;
								take = par != 0;
							}
							if (take) { bestD[u] = D; beste[u] = e; }
						}
					}
				}
#pragma unroll
				for (int u = 0; u < RES_OPT; ++u) {
					if (valid[u]) bufQ[l_out[u]] = bestD[u];
#pragma unroll
					for (int q = 0; q < RES_EMAX; ++q) {
						if ((uint32_t)q < ebits) {
							const unsigned long long word = __ballot(valid[u] && ((beste[u] >> q) & 1u));
							if ((tid & 63u) == 0 && valid[u]) planes[q * nw + (l_out[u] >> 6)] = word;
						}
					}
				}
			}
		}
		const unsigned long long tq2 = DBG ? __builtin_readcyclecounter() : 0ull;
		__syncthreads();
		uint32_t* tmp = bufP; bufP = bufQ; bufQ = tmp;
		if (DBG) { const unsigned long long tq3 = __builtin_readcyclecounter(); acc_a += tq1 - tq0; acc_b += tq2 - tq1; acc_c += tq3 - tq2; nsteps_dbg++; }
	}
	const unsigned long long t_cols = DBG ? __builtin_readcyclecounter() : 0ull;
	// exit slice in logical order, and the run's backtrace record [workgroup][stage_words]
	const uint32_t wout = deposit_args(w, sg.out_grid, sg.n_out_grid);
	if (!(DBG && (P.dbg_flags & 1u)))
	for (uint32_t l = tid; l < (1u << sg.Lf_last); l += NT) {
		const uint32_t idx = wout | deposit_args(l, sg.out_local, sg.n_out_local);
		const uint32_t v = bufP[l];
		cur[idx] = v;
		if (SYM && sg.mirror_out) cur[idx ^ sg.out_fullmask] = v;  // halved run whose reader wants every entry
	}
	unsigned long long* rec = reinterpret_cast<unsigned long long*>(P.bt + (((unsigned long long)sg.bt_hi << 32) | sg.bt_lo)) + (size_t)w * sg.stage_words;
	const unsigned long long* st64 = reinterpret_cast<const unsigned long long*>(stage);
	if (!(DBG && (P.dbg_flags & 2u)))
	for (uint32_t i = tid; i < sg.stage_words; i += NT) rec[i] = st64[i];
	if (score_out && w == 0 && tid == 0) *score_out = bufP[0];
	if (DBG && tid == 0 && sg.pad >= 100 && sg.pad < 104) {
		unsigned long long* dw = P.dbg + (size_t)P.dbg_wg_off + ((size_t)(sg.pad - 100) * 512 + w) * 2;
		dw[0] = rt_begin;
		dw[1] = wall_clock64();
	}
	if (DBG && w == 0 && tid == 0) {
		unsigned long long* d = P.dbg + (size_t)sg.pad * 8;
		d[0] = t_ready - t_begin;
		d[1] = t_cols - t_ready;
		d[2] = __builtin_readcyclecounter() - t_cols;
		d[3] = sg.ncols;
		if (P.dbg_flags & 4u) { d[4] = (t_loaded - t_begin) * nsteps_dbg; d[5] = (t_args - t_begin) * nsteps_dbg; d[6] = (t_first - t_begin) * nsteps_dbg; d[7] = nsteps_dbg; }
		else {
		d[4] = acc_a; d[5] = acc_b; d[6] = acc_c; d[7] = nsteps_dbg;
		}
	}
}

template <bool DBG, bool SYM>
__global__ __launch_bounds__(1024) void resident_segment(DevProblem P, ResSegment sg, const uint32_t* __restrict__ prev,
                                                          uint32_t* __restrict__ cur, uint32_t* __restrict__ score_out) {
	const unsigned long long t_begin = DBG ? __builtin_readcyclecounter() : 0ull;
	touch_kernel_arguments<sizeof(DevProblem) + sizeof(ResSegment) + 24>();
	resident_segment_body<DBG, SYM>(P, sg, prev, cur, blockIdx.x, score_out, t_begin);
}

// One launch = the next run of SEVERAL independent jobs (connected components of one table): blockIdx.y selects the
// entry, blockIdx.x the workgroup of that run (workgroups beyond the run's grid leave at once).  A table of many small
// components is otherwise limited by the dispatch rate (~200 k launches/s), not by the work.
template <bool SYM>
__global__ __launch_bounds__(1024) void resident_batch(DevProblem P, const ResBatchEntry* __restrict__ entries) {
	const ResBatchEntry* __restrict__ e = entries + blockIdx.y;
	{   // pull every 64-byte line of the entry into the scalar cache with independent loads (see touch_kernel_arguments)
		const uint32_t* lines = reinterpret_cast<const uint32_t*>(e);
		uint32_t acc = 0;
#pragma unroll
		for (uint32_t l = 0; l < (sizeof(ResBatchEntry) + 63) / 64; ++l) acc |= lines[l * 16];
		asm volatile("" ::"s"(__builtin_amdgcn_readfirstlane(acc)));
	}
	const ResSegment sg = e->sg;
	if (blockIdx.x >= (1u << (sg.g - sg.half))) return;
	resident_segment_body<false, SYM>(P, sg, e->prev, e->cur, blockIdx.x, e->score_out, 0ull);
}
