// kernels_pedslots.h -- pedigree slot runs (slots.h): pedslot_tables (once per table) and pedslot_run (one launch per run).
// Included by dp_device.hip inside namespace whamd { namespace { ... } }: not a stand-alone header.
//
// Restates compute_column of the reference (src/pedigreedptable.cpp:177-335) for T = 4 / 16 transmission values:
//   D_c[x][i] = cost_{c,i}(x) (+) min_j ( Pr_{c-1}[x & lowmask][j] + popcount(i ^ j) * recomb_c ),  lowest j on ties (:264-300)
//   Pr_c[y][i] = min over the cells x that project onto y, first in Gray-code order on ties (:306-327)
// A lane holds ONE value: workgroup w, thread tid <-> cell (w << L) | (tid >> TB), transmission value tid & (T - 1).
//   * cost: min over NF forms of A[wave][c][t][f] + S[c][lane][f] (tables of slots.h; absent forms are INF + 0); NF = PSLOT_FACT: the
//     factorised line of a trio with untrusted genotypes -- three sums and twelve constants, slots.h;
//   * min over j: butterfly over the TB low lane bits with DPP moves -- popcount(i ^ j) * recomb is a sum over the bits, bit s
//     of the butterfly chooses between "j_s = i_s" (the value the lane holds) and "j_s != i_s" (the partner's + recomb); low
//     bits first, and a tie keeps the candidate whose bit s is 0, so the surviving j is the lowest one;
//   * ending reads: as in kernels_slots.h -- cross-lane move (lane slot) or LDS exchange (wave slot), tie rule of slots.h;
//   * record: one byte per lane and column, argj | ending-read decisions << 4, four columns per stored word.

// ---- tables: G [2^g][fwn], W [2^lw][fwn], S [ncols][64][NF] per run; blockIdx.y = run
__global__ __launch_bounds__(256) void pedslot_tables(DevProblem P, const SlotRun* __restrict__ runs, const PedSlotExtra* __restrict__ extras,
                                                       uint32_t* __restrict__ tab, const DevTerm* __restrict__ fterms) {
	const SlotRun& run = runs[blockIdx.y];
	const PedSlotExtra& ex = extras[blockIdx.y];
	const bool fact = ex.nf == (uint32_t)PSLOT_FACT;   // entries of the factorised line (Problem::fterms) instead of cost forms
	const uint32_t TB = ex.tb, T = 1u << TB, NA = pslot_na(ex.nf), NS = pslot_ns(ex.nf), fwn = ex.fwn, L = run.L, nls = 6u - TB;
	const uint32_t n_g = fwn << run.g, n_w = fwn << run.lw, n_s = run.ncols * 64u * NS, n_k = run.ncols * T * pslot_nk(ex.nf);
	uint32_t* __restrict__ out = tab + (((unsigned long long)ex.g_hi << 32) | ex.g_lo);
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_g + n_w + n_s + n_k; i += gridDim.x * blockDim.x) {
		uint32_t kind, unit, c, t, f;
		if (i >= n_g + n_w + n_s) {   // K [c][t][12]: the constants of the factorised line (entries 4 .. 15 of the column's sixteen)
			const uint32_t r = i - n_g - n_w - n_s;
			out[i] = fterms[((size_t)run.c0 * T + r / PSLOT_NK) * 16u + 4u + r % PSLOT_NK].c;
			continue;
		}
		if (i < n_g + n_w) {
			kind = i < n_g ? 0u : 1u;
			const uint32_t r = kind ? i - n_g : i;
			unit = r / fwn;
			const uint32_t q = r % fwn;   // [c][t][f]
			c = q / (T * NA); t = (q / NA) % T; f = q % NA;
		} else {
			kind = 2u;
			const uint32_t r = i - n_g - n_w;   // [c][lane][f]
			c = r / (64u * NS); unit = (r / NS) & 63u; f = r % NS;
			t = unit & (T - 1u);
		}
		const PedSlotRow& row = P.pslot_rows[run.row_off + c];
		const DevColumn& col = P.cols[run.c0 + c];
		const uint32_t q0 = P.term_ptr[col.term_off + t] + f, q1 = P.term_ptr[col.term_off + t + 1];
		uint32_t acc = kind == 0u ? 0xFFFFFFFFu : 0u;   // absent form: INF + 0 + 0
		if (fact || q0 < q1) {
			const DevTerm tm = fact ? fterms[((size_t)(run.c0 + c) * T + t) * 16u + f] : P.terms[q0];
			acc = kind == 0u ? tm.c : 0u;
			uint32_t s0, s1, bits;
			if (kind == 0u) { s0 = L; s1 = L + run.g; bits = unit; }
			else if (kind == 1u) { s0 = nls; s1 = L; bits = unit; }
			else { s0 = 0; s1 = nls; bits = unit >> TB; }
			for (uint32_t s = s0; s < s1; ++s) {
				if (!((bits >> (s - s0)) & 1u)) continue;
				const uint32_t ind = row.ind[s];
				if ((tm.plus >> ind) & 1u) acc += (uint32_t)row.dslot[s];
				else if ((tm.minus >> ind) & 1u) acc -= (uint32_t)row.dslot[s];
			}
		}
		out[i] = acc;   // (W and S follow G at w_off = n_g and s_off = n_g + n_w, K follows S)
	}
}

// value of lane (l ^ X) for X = 1, 2, 4, 8 as DPP moves (no LDS traffic; every lane has a source, so no `old` value is needed)
template <int X>
__device__ __forceinline__ uint32_t pslot_lane_xor(uint32_t v) {
	if (X == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
	if (X == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);          // quad_perm [2,3,0,1]
	if (X == 4) {   // row_half_mirror (i ^ 7) then quad_perm [3,2,1,0] (i ^ 3)
		const int h = __builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);
		return (uint32_t)__builtin_amdgcn_mov_dpp(h, 0x1B, 0xF, 0xF, true);
	}
	// X == 8: row_mirror (i ^ 15) then row_half_mirror (i ^ 7)
	const int h = __builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);
	return (uint32_t)__builtin_amdgcn_mov_dpp(h, 0x141, 0xF, 0xF, true);
}

__device__ __forceinline__ uint32_t pslot_sat_add(uint32_t a, uint32_t b) { return __builtin_elementwise_add_sat(a, b); }

template <int TB, int NF, bool SPEC>
__device__ __forceinline__ void pedslot_run_body(const DevProblem& P, const SlotRun& run, const PedSlotExtra& ex, const uint32_t* __restrict__ prev,
                                                 uint32_t* __restrict__ cur, const uint32_t w) {
	constexpr uint32_t T = 1u << TB;
	constexpr int NLS = 6 - TB;   // lane slots
	constexpr bool FACT = NF == PSLOT_FACT;          // the factorised line of a trio with untrusted genotypes (slots.h)
	constexpr int NA = (int)pslot_na(NF), NS = (int)pslot_ns(NF), NK = (int)pslot_nk(NF);   // words per (column, value) of A, per (column, lane) of S, per (column, value) of K
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t wave = uni(tid >> 6);
	const uint32_t threads = run.threads, ncols = run.ncols, L = run.L;
	const uint32_t t = lane & (T - 1u);
	const uint32_t lcell = tid >> TB;              // local cell index: wave << NLS | lane >> TB
	const uint32_t Pcell = (w << L) | lcell;       // physical cell index
	// LDS: wave-slot exchange 2 x [threads] | hot lines [PSLOT_MAXCOLS + 4][8] | A [waves][arow] | S [ncols + 4][64][NS] | K [ncols + 4][T][NK]
	uint32_t* hot_lds = smem + 2u * threads;
	uint32_t* a_lds = hot_lds + (PSLOT_MAXCOLS + 4) * 8;
	uint32_t* s_lds = a_lds + (threads >> 6) * (ex.arow + 4u * T * NA);
	uint32_t* k_lds = s_lds + (ncols + 4u) * 64u * NS;
	const uint32_t* __restrict__ tabG = P.pslot_tab + (((unsigned long long)ex.g_hi << 32) | ex.g_lo);
	const PedSlotRow* __restrict__ rows = P.pslot_rows + run.row_off;

	// ---- prologue: one batch of global loads, issued before anything waits
	// (1) the hot lines: 2 x 16 bytes per column
	uint4 hot_piece = make_uint4(0, 0, 0, 0);
	if (tid < ncols * 2u) hot_piece = reinterpret_cast<const uint4*>(rows + (tid >> 1))[tid & 1u];
	// (2) A = G[w] + W[wave]: every wave its own row
	// (AB batches of 64 words requested here; what does not fit is copied by the loop below, whose every trip is load - wait - store: a quartet's
	// 300 words were five L2 round trips in a row -- 1.26 -> 1.41 M columns/s with six batches.  A trio with trusted genotypes has 104 words:
	// two batches, and four measured 7 % slower there.)
	const uint32_t fwn = ex.fwn;
	constexpr int AB = (TB == 2 && NF == 2) ? 2 : 6;
	uint32_t ga[AB], wa[AB];
#pragma unroll
	for (int u = 0; u < AB; ++u) {
		const uint32_t i = (uint32_t)u * 64u + lane;
		ga[u] = 0; wa[u] = 0;
		if (i < fwn) { ga[u] = tabG[(size_t)w * fwn + i]; wa[u] = tabG[ex.w_off + wave * fwn + i]; }
	}
	// (3) S: the same for every workgroup
	const uint4* __restrict__ tabS = reinterpret_cast<const uint4*>(tabG + ex.s_off);
	const uint32_t n_s4 = ncols * 16u * NS;   // ncols * 64 * NS / 4
	uint4 sp[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
#pragma unroll
	for (int u = 0; u < 2; ++u) {
		const uint32_t i = (uint32_t)u * threads + tid;
		if (i < n_s4) sp[u] = tabS[i];
	}
	// (3b) K: the constants of the factorised lines, the same for every workgroup and wave: at most 32 * 4 * 12 words = 384 16-byte pieces -- one per
	// thread of a full workgroup, up to six per thread of a 64-thread one (small tables)
	constexpr int KB = FACT ? 6 : 1;
	const uint32_t n_k4 = FACT ? ncols * T * (NK / 4) : 0u;
	uint4 kp[KB];
#pragma unroll
	for (int u = 0; u < KB; ++u) {
		const uint32_t i = (uint32_t)u * threads + tid;
		kp[u] = make_uint4(0, 0, 0, 0);
		if (i < n_k4) kp[u] = reinterpret_cast<const uint4*>(tabG + ex.s_off + ncols * 64u * NS)[i];
	}
	// (4) the entering value
	uint32_t D = 0;
	if (run.has_prev) {
		const uint32_t occ = run.in_occ;
		uint32_t idx;
		if (run.in_identity) idx = Pcell & occ;
		else {
			idx = 0;
#pragma unroll
			for (int s = 0; s < SLOT_MAXSLOTS; ++s) idx |= (((Pcell & occ) >> s) & 1u) << slot_pos_dev(run.in_pos, s);
		}
		D = prev[(size_t)idx * T + t];
	}
	if (tid < ncols * 2u) reinterpret_cast<uint4*>(hot_lds)[tid] = hot_piece;
#pragma unroll
	for (int u = 0; u < KB; ++u) {
		const uint32_t i = (uint32_t)u * threads + tid;
		if (i < n_k4) reinterpret_cast<uint4*>(k_lds)[i] = kp[u];
	}
	uint32_t* a_row = a_lds + wave * (ex.arow + 4u * T * NA);
#pragma unroll
	for (int u = 0; u < AB; ++u) {
		const uint32_t i = (uint32_t)u * 64u + lane;
		if (i < fwn) a_row[i] = ga[u] + wa[u];
	}
	for (uint32_t i = (uint32_t)AB * 64u + lane; i < fwn; i += 64u) a_row[i] = tabG[(size_t)w * fwn + i] + tabG[ex.w_off + wave * fwn + i];   // long runs / many forms
#pragma unroll
	for (int u = 0; u < 2; ++u) {
		const uint32_t i = (uint32_t)u * threads + tid;
		if (i < n_s4) reinterpret_cast<uint4*>(s_lds)[i] = sp[u];
	}
	for (uint32_t i = 2u * threads + tid; i < n_s4; i += threads) reinterpret_cast<uint4*>(s_lds)[i] = tabS[i];
	// one control byte per column (PSLOT_MAXCOLS = 32): the eight words sit in the lanes of ONE vector register, a trip fetches its word
	// with a v_readlane (as a queue of SGPRs rotated by scalar moves it cost nine instructions per trip, kernels_slots.h)
	const uint32_t ctrl_v = P.slot_ctrl[run.ctrl_off + (lane & 7u)];
	uint32_t* __restrict__ rec = reinterpret_cast<uint32_t*>(P.bt + (((unsigned long long)run.rec_hi << 32) | run.rec_lo)) + (size_t)w * ex.rec_words + tid;
	uint32_t xsel = 0;
	uint32_t tbit[TB > 0 ? TB : 1];
#pragma unroll
	for (int s = 0; s < TB; ++s) tbit[s] = (t >> s) & 1u;
	__syncthreads();

	// What a column needs from LDS, requested ahead (LDS returns in order): {recomb, M0} of the hot line (wave-uniform words in
	// VECTOR registers, see kernels_slots.h), the lane's NF entries of A and of S.
	struct Line { uint2 h; uint32_t a[NA]; uint32_t s[NS]; uint32_t k[NK > 0 ? NK : 1]; };
	const uint32_t a_base = wave * (ex.arow + 4u * T * NA) + t * NA;
	// (lines are requested in column order: three running word offsets advance by a constant per request -- kernels_slots.h; the one of
	// the hot line starts from an opaque move so that the compiler does not learn that its loads are wave-uniform)
	uint32_t hot_at = 0, a_at = a_base, s_at = lane * NS, k_at = t * NK;
	asm volatile("" : "+v"(hot_at));
	auto load_line = [&](uint32_t) -> Line {   // (the argument documents which column a call site requests: always the next one)
		Line ln;
		ln.h = *reinterpret_cast<const uint2*>(hot_lds + hot_at);
		const uint32_t* ap = a_lds + a_at;
		const uint32_t* spn = s_lds + s_at;
		hot_at += 8u;
		a_at += T * NA;
		s_at += 64u * NS;
		if constexpr (FACT) {   // one 16-byte read of A, one of S, three of K
			const uint4 av = *reinterpret_cast<const uint4*>(ap), sv = *reinterpret_cast<const uint4*>(spn);
			ln.a[0] = av.x; ln.a[1] = av.y; ln.a[2] = av.z;
			ln.s[0] = sv.x; ln.s[1] = sv.y; ln.s[2] = sv.z;
			const uint4* kq = reinterpret_cast<const uint4*>(k_lds + k_at);
			k_at += T * NK;
#pragma unroll
			for (int q = 0; q < 3; ++q) { const uint4 kv = kq[q]; ln.k[4 * q] = kv.x; ln.k[4 * q + 1] = kv.y; ln.k[4 * q + 2] = kv.z; ln.k[4 * q + 3] = kv.w; }
		} else if (NF == 2) {
			const uint2 av = *reinterpret_cast<const uint2*>(ap), sv = *reinterpret_cast<const uint2*>(spn);
			ln.a[0] = av.x; ln.a[1] = av.y; ln.s[0] = sv.x; ln.s[1] = sv.y;
		} else {
#pragma unroll
			for (int q = 0; q < NF / 4; ++q) {   // NF = 4: one 16-byte read each; NF = 16: four
				const uint4 av = reinterpret_cast<const uint4*>(ap)[q], sv = reinterpret_cast<const uint4*>(spn)[q];
				ln.a[4 * q] = av.x; ln.a[4 * q + 1] = av.y; ln.a[NF > 2 ? 4 * q + 2 : 0] = av.z; ln.a[NF > 2 ? 4 * q + 3 : 0] = av.w;
				ln.s[4 * q] = sv.x; ln.s[4 * q + 1] = sv.y; ln.s[NF > 2 ? 4 * q + 2 : 0] = sv.z; ln.s[NF > 2 ? 4 * q + 3 : 0] = sv.w;
			}
		}
		return ln;
	};
	uint32_t recacc = 0;
	auto column = [&](const Line& ln, const uint32_t ci, const uint32_t ctrl, const int sub) {
		const uint32_t rc = ln.h.x;
		// cost of this lane's (cell, transmission value)
		uint32_t cost;
		if constexpr (FACT) {   // (slots.h: the minimum over the sixteen allele assignments, the untransmitted alleles first; 19 operations)
			const uint32_t X = ln.a[0] + ln.s[0], Y = ln.a[1] + ln.s[1], C = ln.a[2] + ln.s[2];
			const uint32_t M0 = min(ln.k[0], ln.k[1] + X), M1 = min(ln.k[2], ln.k[3] - X);
			const uint32_t F0 = min(ln.k[4], ln.k[5] + Y), F1 = min(ln.k[6], ln.k[7] - Y);
			const uint32_t t00 = ln.k[8] + M0 + F0, t01 = ln.k[9] + C + M0 + F1;
			const uint32_t t10 = ln.k[10] - C + M1 + F0, t11 = ln.k[11] + M1 + F1;
			cost = min(min(t00, t01), min(t10, t11));
		} else {
			cost = ln.a[0] + ln.s[0];
#pragma unroll
			for (int f = 1; f < NF; ++f) cost = min(cost, ln.a[f] + ln.s[f]);
		}
		// min over the previous transmission value j, argmin = lowest j
		uint32_t v = D, j = t;
		if (TB >= 1) { const uint32_t pv = pslot_lane_xor<1>(v), pj = pslot_lane_xor<1>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[0]); v = take ? cand : v; j = take ? pj : j; }
		if (TB >= 2) { const uint32_t pv = pslot_lane_xor<2>(v), pj = pslot_lane_xor<2>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[TB >= 2 ? 1 : 0]); v = take ? cand : v; j = take ? pj : j; }
		if (TB >= 3) { const uint32_t pv = pslot_lane_xor<4>(v), pj = pslot_lane_xor<4>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[TB >= 3 ? 2 : 0]); v = take ? cand : v; j = take ? pj : j; }
		if (TB >= 4) { const uint32_t pv = pslot_lane_xor<8>(v), pj = pslot_lane_xor<8>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[TB >= 4 ? 3 : 0]); v = take ? cand : v; j = take ? pj : j; }
		D = pslot_sat_add(v, cost);
		uint32_t byte = j;
		const uint32_t n_end = ctrl & 3u;
		auto ending = [&](const uint32_t M, const uint32_t slot, const uint32_t e) {
			const uint32_t q = (uint32_t)__popc(Pcell & M) & 1u;
			uint32_t other;
			if (slot < (uint32_t)NLS) {
				other = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ (1u << (slot + TB))) << 2), (int)D);
			} else {
				// the partner is another wave's lane: exchange through LDS (two buffers: the barrier of the next exchange also
				// protects this one's reads)
				uint32_t* xb = smem + xsel * threads;
				xb[tid] = D;
				__syncthreads();
				other = xb[tid ^ (64u << (slot - NLS))];
				xsel ^= 1u;
			}
			byte |= (other < pslot_sat_add(D, q) ? 16u : 0u) << e;   // side-0 lane: the pair's decision (the side-1 lane's bit is never read)
			D = min(D, other);
		};
		if (n_end) {
			ending(ln.h.y, (ctrl >> 2) & 31u, 0u);
			if (n_end > 1u) {   // several reads ending in one column (rare): slots out of the hot line -- a VALU -> SGPR transfer
				uint32_t off1 = ci * 8u + 2u;   // hot words 2, 3: info1, M1; words 4, 5: info2, M2
				asm volatile("" : "+v"(off1));
				const uint2 e1 = *reinterpret_cast<const uint2*>(hot_lds + off1);
				uint32_t s1;
				asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1" : "=s"(s1) : "v"(e1.x));
				ending(e1.y, s1 & 255u, 1u);
				if (n_end > 2u) {
					uint32_t off2 = ci * 8u + 4u;
					asm volatile("" : "+v"(off2));
					const uint2 e2 = *reinterpret_cast<const uint2*>(hot_lds + off2);
					uint32_t s2;
					asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1" : "=s"(s2) : "v"(e2.x));
					ending(e2.y, s2 & 255u, 2u);
					// a control byte of 3 says "three or more": the fourth (PSLOT_MAXEND) comes out of the row itself
					const PedSlotRow* __restrict__ grow = P.pslot_rows + run.row_off + ci;
					const uint32_t total = *(const __attribute__((address_space(4))) uint32_t*)(unsigned long long)(&grow->n_end);
					if (total > 3u) {
						const uint32_t s3 = *(const __attribute__((address_space(4))) uint32_t*)(unsigned long long)(&grow->pad[2]);
						const uint32_t m3 = *(const __attribute__((address_space(4))) uint32_t*)(unsigned long long)(&grow->pad[3]);
						ending(m3, s3 & 255u, 3u);
					}
				}
			}
		}
		recacc |= byte << (8 * sub);
	};
	{
		// four columns per trip (one control word, one record word); every line is requested three columns ahead
		Line h0 = load_line(0), h1 = load_line(1), h2 = load_line(2), h3;
		for (uint32_t ci = 0; ci < ncols; ci += 4u) {
			const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)ctrl_v, (int)(ci >> 2));
			recacc = 0;
			h3 = load_line(ci + 3u);      // (lines beyond the run may be read: the LDS areas have room, the values are not used)
			column(h0, ci, cw & 255u, 0);
			if (ci + 1u < ncols) {
				h0 = load_line(ci + 4u);
				column(h1, ci + 1u, (cw >> 8) & 255u, 1);
				if (ci + 2u < ncols) {
					h1 = load_line(ci + 5u);
					column(h2, ci + 2u, (cw >> 16) & 255u, 2);
					if (ci + 3u < ncols) {
						h2 = load_line(ci + 6u);
						column(h3, ci + 3u, cw >> 24, 3);
					}
				}
			}
			rec[(size_t)(ci >> 2) * threads] = recacc;   // fire and forget
		}
	}

	// ---- exit: scatter into the next step's order (lanes whose free-slot bits are zero hold the representatives)
	{
		const uint32_t occ = run.out_occ;
		const uint32_t localmask = (1u << L) - 1u;
		const bool writes = (lcell & ~occ & localmask) == 0u;
		uint32_t idx = 0;
#pragma unroll
		for (int s = 0; s < SLOT_MAXSLOTS; ++s) idx |= (((Pcell & occ) >> s) & 1u) << slot_pos_dev(run.out_pos, s);
		unsigned long long best_key = ~0ull;
		if (writes) {
			cur[(size_t)idx * T + t] = D;
			if (SPEC) best_key = ((unsigned long long)D << 32) | (idx * T + t);
		}
		if (SPEC && run.spec_id) {
			// seed of the speculative backtrace (kernels_backtrace.h): the smallest entry (value, exit index * T + t) this wave stored
#pragma unroll
			for (int m = 1; m < 64; m <<= 1) {
				const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best_key, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best_key >> 32), m);
				best_key = min(best_key, ((unsigned long long)hi << 32) | lo);
			}
			if (lane == 0) P.spec_keys[(size_t)(run.spec_id - 1u) * P.spec_stride + w * (threads >> 6) + wave] = best_key;
		}
	}
}

template <int TB, int NF, bool SPEC>
__global__ __launch_bounds__(512) void pedslot_run(DevProblem P, SlotRun run, PedSlotExtra ex, const uint32_t* __restrict__ prev,
                                                   uint32_t* __restrict__ cur) {
	touch_kernel_arguments<sizeof(DevProblem) + sizeof(SlotRun) + sizeof(PedSlotExtra) + 16>();
	pedslot_run_body<TB, NF, SPEC>(P, run, ex, prev, cur, blockIdx.x);
}

// One launch = the next run of SEVERAL pedigree tables (see slot_group, kernels_slots.h): blockIdx.y selects the table's entry.
template <int TB, int NF>
__global__ __launch_bounds__(512, NF == 2 ? 8 : (NF == 4 ? 4 : 2)) void pedslot_group(SlotGroupArgs args) {   // (NF = 2: four workgroups per CU -- at most 80 SGPRs, 64 VGPRs)
	const SlotGroupWho who = slot_group_who(args);   // (the table as the fast grid dimension: a table's workgroups on one XCD, kernels_slots.h)
	if (who.none) return;
	const uint32_t warm = slot_warm_next(args.entry[who.table]);   // (the next step's entry into this XCD's L2: kernels_slots.h)
	const SlotBatchEntry e = slot_scalar_copy(args.entry[who.table]);
	const SlotRun& run = e.run;
	if (who.w >= (1u << run.g) || threadIdx.x >= run.threads) return;
	const DevProblem P = slot_entry_problem(e, true);
	if (run.spec_id) pedslot_run_body<TB, NF, true>(P, run, e.ex, e.prev, e.cur, who.w);
	else pedslot_run_body<TB, NF, false>(P, run, e.ex, e.prev, e.cur, who.w);
	slot_warm_done(warm);
}
