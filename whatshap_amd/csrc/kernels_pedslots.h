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
	const bool fact4 = ex.nf == (uint32_t)PSLOT_FACT4;   // a quartet's line: per COLUMN (twenty entries), the same for every transmission value
	const bool fact = ex.nf == (uint32_t)PSLOT_FACT || fact4;   // entries of the factorised line (Problem::fterms) instead of cost forms
	const uint32_t TB = ex.tb, T = 1u << TB, TA = pslot_ta(ex.nf, T), NA = pslot_na(ex.nf), NS = pslot_ns(ex.nf), fwn = ex.fwn, L = run.L, nls = 6u - TB;
	const uint32_t n_g = fwn << run.g, n_w = fwn << run.lw, n_s = run.ncols * 64u * NS, n_k = run.ncols * T * pslot_nk(ex.nf);
	uint32_t* __restrict__ out = tab + (((unsigned long long)ex.g_hi << 32) | ex.g_lo);
	// X runs (pedslot_runx_body): recombination cost and control word per column, the lanes' and the workgroups' tie parities (slots.h PedSlotExtra::x_off)
	const uint32_t n_x = (run.yflags & 8u) ? (uint32_t)pslotx_words(run.ncols, run.threads, run.g) : 0u;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_x; i += gridDim.x * blockDim.x) {
		const uint32_t ncp = run.ncols + (uint32_t)SLOT_XPAD;
		const PedSlotRow* __restrict__ rows = P.pslot_rows + run.row_off;
		auto end_info = [&](const PedSlotRow& row, uint32_t k, uint32_t& slot, uint32_t& M) {
			if (k == 0) { slot = row.info0 & 255u; M = row.M0; }
			else if (k == 1) { slot = row.info1 & 255u; M = row.M1; }
			else if (k == 2) { slot = row.info2 & 255u; M = row.M2; }
			else { slot = row.pad[2] & 255u; M = row.pad[3]; }
		};
		uint32_t word = 0;
		if (i < ncp) word = i < run.ncols ? rows[i].recomb : 0xFFFFFFFFu;
		else if (i < 2u * ncp) {
			const uint32_t c = i - ncp;
			if (c < run.ncols) {
				uint32_t exchanges = 0;   // wave-slot endings of the run before this column: they alternate between two LDS buffers
				for (uint32_t c2 = 0; c2 < c; ++c2)
					for (uint32_t k = 0; k < rows[c2].n_end; ++k) { uint32_t sl, M; end_info(rows[c2], k, sl, M); exchanges += sl >= nls; }
				word = rows[c].n_end;
				for (uint32_t k = 0; k < rows[c].n_end; ++k) {
					uint32_t sl, M;
					end_info(rows[c], k, sl, M);
					word |= (sl | ((sl >= nls ? exchanges & 1u : 0u) << 3)) << (3u + 4u * k);
					exchanges += sl >= nls;
				}
			}
		} else {
			const uint32_t q = i - 2u * ncp;
			const uint32_t index = q < run.threads ? (q >> TB) : ((q - run.threads) << L);   // a lane's local cell index / a workgroup's grid bits
			const uint32_t keep = q < run.threads ? (1u << L) - 1u : ~((1u << L) - 1u);
			uint32_t e = 0;
			for (uint32_t c = 0; c < run.ncols; ++c)
				for (uint32_t k = 0; k < rows[c].n_end; ++k, ++e) { uint32_t sl, M; end_info(rows[c], k, sl, M); word |= ((uint32_t)__popc(index & keep & M) & 1u) << (e & 31u); }
		}
		out[ex.x_off + i] = word;
	}
	// ---- G [unit][c][t][f] and W: ONE thread per (column, value, form) walks the 2^g workgroups (2^lw waves) in Gray order -- every step flips one slot: one add.
	// (One thread per WORD -- five runtime divisions, a dozen scattered loads and a loop over the set bits each -- built configs[3]'s 0.9 GB in 3.7 ms, 243 GB/s,
	//  exposed between the end of a create and the first launch of a fresh table; u32 sums are modular: the order of the additions does not matter.)
	__shared__ uint32_t sd[SLOT_GMAX + 1][256];   // [slot of the unit's bits][thread]: the signed delta the form takes from that slot (0: none)
	for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < 2u * fwn; item += gridDim.x * blockDim.x) {
		const uint32_t kind = item >= fwn ? 1u : 0u, q = item - kind * fwn;   // q: [c][t][f]
		const uint32_t c = q / (TA * NA), t = (q / NA) % TA, f = q % NA;
		const PedSlotRow& row = P.pslot_rows[run.row_off + c];
		const DevColumn& col = P.cols[run.c0 + c];
		const uint32_t q0 = P.term_ptr[col.term_off + t] + f, q1 = P.term_ptr[col.term_off + t + 1];
		const bool present = fact || q0 < q1;
		DevTerm tm{};
		if (present) tm = fact4 ? fterms[(size_t)(run.c0 + c) * PSLOT_FSTRIDE4 + f] : (fact ? fterms[((size_t)(run.c0 + c) * T + t) * 16u + f] : P.terms[q0]);
		const uint32_t s0 = kind ? nls : L, nb = kind ? L - nls : run.g;
		uint32_t acc = kind == 0u ? (present ? tm.c : 0xFFFFFFFFu) : 0u;   // absent form: INF + 0 + 0
		for (uint32_t b = 0; b < nb && b <= (uint32_t)SLOT_GMAX; ++b) {
			uint32_t d = 0;
			if (present) {
				const uint32_t ind = row.ind[s0 + b];
				if ((tm.plus >> ind) & 1u) d = (uint32_t)row.dslot[s0 + b];
				else if ((tm.minus >> ind) & 1u) d = 0u - (uint32_t)row.dslot[s0 + b];
			}
			sd[b][threadIdx.x] = d;
		}
		uint32_t* __restrict__ dst = out + (kind ? n_g : 0u) + q;
		dst[0] = acc;
		uint32_t gray = 0;
		for (uint32_t u = 1; u < (1u << nb); ++u) {
			const uint32_t b = (uint32_t)__builtin_ctz(u);
			gray ^= 1u << b;
			const uint32_t d = sd[b][threadIdx.x];
			acc = ((gray >> b) & 1u) ? acc + d : acc - d;
			dst[(size_t)gray * fwn] = acc;
		}
	}
	for (uint32_t i = n_g + n_w + blockIdx.x * blockDim.x + threadIdx.x; i < n_g + n_w + n_s + n_k; i += gridDim.x * blockDim.x) {
		uint32_t kind, unit, c, t, f;
		if (i >= n_g + n_w + n_s) {   // K [c][t][12]: the constants of the factorised line (entries 4 .. 15 of the column's sixteen)
			const uint32_t r = i - n_g - n_w - n_s;
			out[i] = fact4 ? fterms[((size_t)run.c0 + r / PSLOT_NK4) * PSLOT_FSTRIDE4 + 4u + r % PSLOT_NK4].c
			               : fterms[((size_t)run.c0 * T + r / PSLOT_NK) * 16u + 4u + r % PSLOT_NK].c;
			continue;
		}
		{   // S [c][lane][f]: the lane-slot part of a form, per lane
			kind = 2u;
			const uint32_t r = i - n_g - n_w;
			c = r / (64u * NS); unit = (r / NS) & 63u; f = r % NS;
			t = unit & (T - 1u);
		}
		const PedSlotRow& row = P.pslot_rows[run.row_off + c];
		const DevColumn& col = P.cols[run.c0 + c];
		const uint32_t q0 = P.term_ptr[col.term_off + t] + f, q1 = P.term_ptr[col.term_off + t + 1];
		uint32_t acc = kind == 0u ? 0xFFFFFFFFu : 0u;   // absent form: INF + 0 + 0
		if (fact || q0 < q1) {
			const DevTerm tm = fact4 ? fterms[(size_t)(run.c0 + c) * PSLOT_FSTRIDE4 + f] : (fact ? fterms[((size_t)(run.c0 + c) * T + t) * 16u + f] : P.terms[q0]);
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
	// X == 8: row_ror:8 -- a rotation by half a row of sixteen lanes is i ^ 8 (one move; row_mirror then row_half_mirror were two)
	return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);
}

__device__ __forceinline__ uint32_t pslot_sat_add(uint32_t a, uint32_t b) { return __builtin_elementwise_add_sat(a, b); }

template <int TB, int NF, bool SPEC, bool PACKED>
__device__ __forceinline__ void pedslot_run_body(const DevProblem& P, const SlotRun& run, const PedSlotExtra& ex, const uint32_t* __restrict__ prev,
                                                 uint32_t* __restrict__ cur, const uint32_t w) {
	constexpr uint32_t T = 1u << TB;
	constexpr int NLS = 6 - TB;   // lane slots
	constexpr bool FACT = NF == PSLOT_FACT;          // the factorised line of a trio with untrusted genotypes (slots.h)
	constexpr bool FACT4 = NF == PSLOT_FACT4;        // ... of a quartet: A and K per column (the same for every transmission value), sixteen constants
	static_assert(!FACT4 || TB == 4, "PSLOT_FACT4 is the line of a quartet");
	constexpr uint32_t TA = FACT4 ? 1u : T;          // values per column in A (and K)
	constexpr int KL = FACT4 ? (int)PSLOT_NK4 : (NF == PSLOT_FACT ? (int)PSLOT_NK : 1);   // constants a lane reads per column
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
	constexpr int AB = ((TB == 2 && NF == 2) || FACT4) ? 2 : 6;
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
	const uint32_t n_k4 = FACT ? ncols * T * (NK / 4) : (FACT4 ? ncols * (PSLOT_NK4 / 4) : 0u);   // (a quartet's: at most 128 pieces, the loop below takes what a 64-thread workgroup leaves)
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
	if (FACT4) for (uint32_t i = (uint32_t)KB * threads + tid; i < n_k4; i += threads) reinterpret_cast<uint4*>(k_lds)[i] = reinterpret_cast<const uint4*>(tabG + ex.s_off + ncols * 64u * NS)[i];
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
	struct Line { uint2 h; uint32_t a[NA]; uint32_t s[NS]; uint32_t k[KL]; };
	const uint32_t a_base = wave * (ex.arow + 4u * T * NA) + (FACT4 ? 0u : t * NA);
	// a quartet's factorised line: how this lane's transmission value wires the children to the founders' haplotypes (slots.h, Problem::fact4_roles)
	const bool r_u1 = FACT4 && ((ex.pad[0] >> t) & 1u), r_v1 = FACT4 && ((ex.pad[0] >> (16u + t)) & 1u);
	const bool r_su = FACT4 && ((ex.pad[1] >> t) & 1u), r_sv = FACT4 && ((ex.pad[1] >> (16u + t)) & 1u);
	// (lines are requested in column order: three running word offsets advance by a constant per request -- kernels_slots.h; the one of
	// the hot line starts from an opaque move so that the compiler does not learn that its loads are wave-uniform)
	uint32_t hot_at = 0, a_at = a_base, s_at = lane * NS, k_at = FACT4 ? 0u : t * NK;
	asm volatile("" : "+v"(hot_at));
	auto load_line = [&](uint32_t) -> Line {   // (the argument documents which column a call site requests: always the next one)
		Line ln;
		ln.h = *reinterpret_cast<const uint2*>(hot_lds + hot_at);
		const uint32_t* ap = a_lds + a_at;
		const uint32_t* spn = s_lds + s_at;
		hot_at += 8u;
		a_at += TA * NA;
		s_at += 64u * NS;
		if constexpr (FACT) {   // one 16-byte read of A, one of S, three of K
			const uint4 av = *reinterpret_cast<const uint4*>(ap), sv = *reinterpret_cast<const uint4*>(spn);
			ln.a[0] = av.x; ln.a[1] = av.y; ln.a[2] = av.z;
			ln.s[0] = sv.x; ln.s[1] = sv.y; ln.s[2] = sv.z;
			const uint4* kq = reinterpret_cast<const uint4*>(k_lds + k_at);
			k_at += T * NK;
#pragma unroll
			for (int q = 0; q < 3; ++q) { const uint4 kv = kq[q]; ln.k[4 * q] = kv.x; ln.k[4 * q + 1] = kv.y; ln.k[4 * q + 2] = kv.z; ln.k[4 * q + 3] = kv.w; }
		} else if constexpr (FACT4) {   // one 16-byte read of A, one of S, four of K (the same addresses for the sixteen lanes of a cell / for every lane)
			const uint4 av = *reinterpret_cast<const uint4*>(ap), sv = *reinterpret_cast<const uint4*>(spn);
			ln.a[0] = av.x; ln.a[1] = av.y; ln.a[2] = av.z; ln.a[3] = av.w;
			ln.s[0] = sv.x; ln.s[1] = sv.y; ln.s[2] = sv.z; ln.s[3] = sv.w;
			const uint4* kq = reinterpret_cast<const uint4*>(k_lds + k_at);
			k_at += PSLOT_NK4;
#pragma unroll
			for (int q = 0; q < 4; ++q) { const uint4 kv = kq[q]; ln.k[4 * q] = kv.x; ln.k[4 * q + 1] = kv.y; ln.k[4 * q + 2] = kv.z; ln.k[4 * q + 3] = kv.w; }
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
	constexpr bool packed = PACKED;   // (SlotRun::yflags bit 4, slot_plan.cpp: every finite value of the table below 2^(31 - TB) - 1; chosen at the launch)
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
		} else if constexpr (FACT4) {   // (slots.h: the minimum over the sixteen allele assignments by elimination)
			cost = pslot_fact4_cost(ln.a[0] + ln.s[0], ln.a[1] + ln.s[1], ln.a[2] + ln.s[2], ln.a[3] + ln.s[3], ln.k, r_u1, r_v1, r_su, r_sv);
		} else {
			cost = ln.a[0] + ln.s[0];
#pragma unroll
			for (int f = 1; f < NF; ++f) cost = min(cost, ln.a[f] + ln.s[f]);
		}
		// min over the previous transmission value j, argmin = lowest j
		uint32_t v = D, j = t;
		if constexpr (packed) {
			// (value, j) as ONE key, value << TB | j: the minimum of two keys is the smaller value and, on ties, the lower j -- the rule of the staged
			// comparison below, and a minimum is a minimum in any order.  A stage is the partner's key + the scaled recombination cost (the DPP move folds
			// into the add) and a v_min: 2 instructions instead of 7 -- 9.  Nothing can wrap: the planner sets the flag when every finite value is below
			// CL = 2^(31 - TB) - 1 (Problem::value_bound; it bounds 2 * triples * recomb of every column too), values at or above CL mean INFINITE and are
			// clamped to CL when the key is formed -- keys stay below 2^31, key + cost below 2^32.
			constexpr uint32_t CL = (1u << (31 - TB)) - 1u;
			const uint32_t rcs = rc << TB;
			uint32_t key = (min(D, CL) << TB) | t;
			if (TB >= 1) key = min(key, pslot_lane_xor<1>(key) + rcs);
			if (TB >= 2) key = min(key, pslot_lane_xor<2>(key) + rcs);
			if (TB >= 3) key = min(key, pslot_lane_xor<4>(key) + rcs);
			if (TB >= 4) key = min(key, pslot_lane_xor<8>(key) + rcs);
			v = key >> TB;
			j = key & (T - 1u);
		} else {
		if (TB >= 1) { const uint32_t pv = pslot_lane_xor<1>(v), pj = pslot_lane_xor<1>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[0]); v = take ? cand : v; j = take ? pj : j; }
		if (TB >= 2) { const uint32_t pv = pslot_lane_xor<2>(v), pj = pslot_lane_xor<2>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[TB >= 2 ? 1 : 0]); v = take ? cand : v; j = take ? pj : j; }
		if (TB >= 3) { const uint32_t pv = pslot_lane_xor<4>(v), pj = pslot_lane_xor<4>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[TB >= 3 ? 2 : 0]); v = take ? cand : v; j = take ? pj : j; }
		if (TB >= 4) { const uint32_t pv = pslot_lane_xor<8>(v), pj = pslot_lane_xor<8>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[TB >= 4 ? 3 : 0]); v = take ? cand : v; j = take ? pj : j; }
		}
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
		if (packed && D >= (1u << (31 - TB)) - 1u) D = 0xFFFFFFFFu;   // (the keys' "infinite" back to the value every other kernel tests for)
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

#ifdef WHAMD_DEBUG_BUILD
// ---- X runs for pedigrees (round 5; the single individual's are in kernels_slots.h, where the reasons are) ---------------------------------------------
// NOT in the product library: correct (bit-identical solutions on trio, quartet and the factorised line) but slower than pedslot_run below -- 8.4 against
// 6.1 us per launch on a trio at coverage 15, 12.7 against 7.6 with untrusted genotypes (scripts/gpu_pedx_ab.py with WHAMD_PED_XRUN=1 in the debug library).
// A pedigree column is the butterfly (16 - 40 DPP operations) whichever way its cost arrives, and forming the costs per thread is ~100 four-byte loads.
// The cost of a lane's (cell, transmission value) in a column does not depend on the column before: the prologue forms the costs of EVERY column of the
// run straight from the tables (whatever the number of forms: NF = 2, 4, 16 or the factorised line -- the same loop below for all of them) and keeps them
// in the thread's own 16 bytes per trip of an LDS area (no staging of tables in LDS, no barrier).  The loop -- counted, over pairs of trips of four columns --
// fetches per trip one such line, four recombination costs and four control words through the scalar cache; a column is the butterfly over the previous
// transmission value, one saturating add and the record byte.  Columns behind the run's last are harmless: their recombination cost is all-ones (no
// candidate of the butterfly ever wins) and their cost is zero.  An ending read is one hand-written block (the partner's value requested first, the
// lane's tie parity a bit of a create-time word); all four ending reads of a column are described in its control word.
#define PSLOTX_ENDING_ASM                                                                                              \
	"s_and_b32 %[sl], %[f], 7\n\t"                 /* slot of the ending read (bit 3 of the field: the exchange buffer) */ \
	"s_cmp_lt_u32 %[sl], %[nls]\n\t"                                                                                   \
	"s_cbranch_scc0 .Lpw%=\n\t"                                                                                        \
	"s_add_u32 %[sa], %[sl], %[tb2]\n\t"           /* lane slot: byte address of lane ^ 2^(slot + TB) = (lane * 4) ^ 2^(slot + TB + 2) */ \
	"s_lshl_b32 %[sa], 1, %[sa]\n\t"                                                                                   \
	"v_xor_b32_e32 %[t], %[sa], %[l4]\n\t"                                                                             \
	"ds_bpermute_b32 %[o], %[t], %[d]\n\t"                                                                             \
	"s_branch .Lpm%=\n"                                                                                                \
	".Lpw%=:\n\t"                                  /* wave slot: partner thread = tid ^ (64 << (slot - NLS)), 4 bytes each */ \
	"s_sub_u32 %[sa], %[sl], %[nls]\n\t"                                                                               \
	"s_lshl_b32 %[sa], 0x100, %[sa]\n\t"                                                                               \
	"s_bfe_u32 %[sl], %[f], 0x10003\n\t"                                                                               \
	"s_mul_i32 %[sl], %[sl], %[xby]\n\t"                                                                               \
	"v_add_u32_e32 %[t], %[sl], %[t4]\n\t"                                                                             \
	"ds_write_b32 %[t], %[d]\n\t"                                                                                      \
	"v_xor_b32_e32 %[t], %[sa], %[t4]\n\t"                                                                             \
	"v_add_u32_e32 %[t], %[sl], %[t]\n\t"                                                                              \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"s_barrier\n\t"                                                                                                    \
	"ds_read_b32 %[o], %[t]\n"                                                                                         \
	".Lpm%=:\n\t"                                                                                                      \
	"v_and_b32_e32 %[q], 1, %[par]\n\t"            /* the lane's tie parity: bit 0 of par; the next ending read's moves down */ \
	"v_lshrrev_b32_e32 %[par], 1, %[par]\n\t"                                                                          \
	"v_add_u32_e64 %[q], %[d], %[q] clamp\n\t"     /* mine + q, saturating */                                          \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"v_cmp_lt_u32_e64 %[m], %[o], %[q]\n\t"        /* other < mine + q: the pair's decision (read at the side-0 lane only) */ \
	"v_min_u32_e32 %[d], %[d], %[o]\n\t"                                                                               \
	"s_nop 0\n\t"                                  /* (gfx950: a VALU result in an SGPR may be read by a VALU two instructions later at the earliest) */ \
	"v_cndmask_b32_e64 %[q], 0, 1, %[m]\n\t"                                                                           \
	"v_lshl_or_b32 %[by], %[q], %[sh], %[by]"

template <int SH, int TB>
__device__ __forceinline__ void pslotx_ending(uint32_t& D, uint32_t& byte, uint32_t& par, const uint32_t field, const uint32_t xbytes, const uint32_t lane4, const uint32_t tid4) {
	unsigned long long m;
	uint32_t tmp, o, q, sa, sl;
	asm volatile(PSLOTX_ENDING_ASM
	             : [d] "+v"(D), [by] "+v"(byte), [par] "+v"(par), [t] "=&v"(tmp), [o] "=&v"(o), [q] "=&v"(q), [sa] "=&s"(sa), [sl] "=&s"(sl), [m] "=&s"(m)
	             : [f] "s"(field), [xby] "s"(xbytes), [l4] "v"(lane4), [t4] "v"(tid4), [nls] "n"(6 - TB), [tb2] "n"(TB + 2), [sh] "n"(SH)
	             : "memory", "scc");
}

template <int TB, int NF, int XC, bool SPEC>
__device__ __forceinline__ void pedslot_runx_body(const DevProblem& P, const SlotRun& run, const PedSlotExtra& ex, const uint32_t* __restrict__ prev,
                                                  uint32_t* __restrict__ cur, const uint32_t w) {
	constexpr uint32_t T = 1u << TB;
	constexpr int NLS = 6 - TB;
	constexpr bool FACT = NF == PSLOT_FACT;
	constexpr int NA = (int)pslot_na(NF), NS = (int)pslot_ns(NF), NK = (int)pslot_nk(NF);
	static_assert(XC % 8 == 0 && XC <= SLOT_XCOLS, "whole pairs of trips of four columns");
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // wave-slot exchange 2 x [threads] | the lanes' cost lines [XC / 4 + 3][threads][4]
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t wave = uni(tid >> 6);
	const uint32_t threads = run.threads, ncols = run.ncols, L = run.L;
	const uint32_t t = lane & (T - 1u);
	const uint32_t lcell = tid >> TB;
	const uint32_t Pcell = (w << L) | lcell;
	const uint32_t* __restrict__ tabG = P.pslot_tab + (((unsigned long long)ex.g_hi << 32) | ex.g_lo);
	const uint32_t fwn = ex.fwn;
	typedef const __attribute__((address_space(4))) uint32_t* cptr1;
	typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));
	typedef const __attribute__((address_space(4))) u32x4c* cptr4;
	// ---- prologue: one batch of loads.  Scalars of the first trip through the scalar cache ...
	const uint32_t* __restrict__ rc_tab = tabG + ex.x_off;                            // recombination cost per column
	const uint32_t* __restrict__ cw_tab = rc_tab + ncols + (uint32_t)SLOT_XPAD;       // control word per column
	const uint32_t* __restrict__ par_tab = cw_tab + ncols + (uint32_t)SLOT_XPAD;      // tie parities: [threads], then [workgroups]
	u32x4c rcA = *(cptr4)(unsigned long long)rc_tab, rcB;
	u32x4c cwA = *(cptr4)(unsigned long long)cw_tab, cwB;
	const uint32_t par_w = *(cptr1)(unsigned long long)(par_tab + threads + w);
	// ... the entering value first (written by the previous launch on other XCDs: the longest latency) ...
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
	const uint32_t par_l = par_tab[tid];
	// ... and the cost entries of every column: workgroup part, wave part, lane part (+ the constants of a factorised line)
	const uint32_t* __restrict__ g_src = tabG + (size_t)w * fwn + t * NA;
	const uint32_t* __restrict__ w_src = tabG + ex.w_off + wave * fwn + t * NA;
	const uint32_t* __restrict__ s_src = tabG + ex.s_off + lane * NS;
	const uint32_t* __restrict__ k_src = tabG + ex.s_off + ncols * 64u * NS + t * NK;
	uint32_t cost[XC];
#pragma unroll
	for (int c = 0; c < XC; ++c) {
		uint32_t a[NA], sv[NS];
#pragma unroll
		for (int f = 0; f < NA; ++f) a[f] = g_src[c * (int)T * NA + f] + w_src[c * (int)T * NA + f];
#pragma unroll
		for (int f = 0; f < NS; ++f) sv[f] = s_src[c * 64 * NS + f];
		uint32_t v;
		if constexpr (FACT) {   // (slots.h: the minimum over the sixteen allele assignments, the untransmitted alleles first; 19 operations)
			uint32_t k[NK > 0 ? NK : 1];
#pragma unroll
			for (int f = 0; f < NK; ++f) k[f] = k_src[c * (int)T * NK + f];
			const uint32_t X = a[0] + sv[0], Y = a[1] + sv[1], C = a[2] + sv[2];
			const uint32_t M0 = min(k[0], k[1] + X), M1 = min(k[2], k[3] - X);
			const uint32_t F0 = min(k[4], k[5] + Y), F1 = min(k[6], k[7] - Y);
			const uint32_t t00 = k[8] + M0 + F0, t01 = k[9] + C + M0 + F1;
			const uint32_t t10 = k[10] - C + M1 + F0, t11 = k[11] + M1 + F1;
			v = min(min(t00, t01), min(t10, t11));
		} else {
			v = a[0] + sv[0];
#pragma unroll
			for (int f = 1; f < NF; ++f) v = min(v, a[f] + sv[f]);
		}
		cost[c] = (uint32_t)c < ncols ? v : 0u;   // (behind the run: the tables that follow were read -- the column must change nothing)
	}
	uint4* __restrict__ xs = reinterpret_cast<uint4*>(smem + 2u * threads) + tid;   // + trip * threads (behind the two exchange buffers)
#pragma unroll
	for (int t4 = 0; t4 < XC / 4; ++t4) xs[(uint32_t)t4 * threads] = make_uint4(cost[4 * t4], cost[4 * t4 + 1], cost[4 * t4 + 2], cost[4 * t4 + 3]);
	uint32_t par = par_l ^ par_w;
	uint32_t* __restrict__ rec = reinterpret_cast<uint32_t*>(P.bt + (((unsigned long long)run.rec_hi << 32) | run.rec_lo)) + (size_t)w * ex.rec_words + tid;
	uint32_t tbit[TB > 0 ? TB : 1];
#pragma unroll
	for (int s = 0; s < TB; ++s) tbit[s] = (t >> s) & 1u;

	const uint32_t lds0 = (uint32_t)(unsigned long long)smem;
	const uint32_t lane4 = lane << 2, tid4 = lds0 + (tid << 2);
	const uint32_t xbytes = threads * 4u;
	// one column: the butterfly over the previous transmission value (lowest j on ties), the cost, the ending reads; returns the record byte
	auto column = [&](const uint32_t cst, const uint32_t rc, const uint32_t ctrl) -> uint32_t {
		uint32_t v = D, j = t;
		if (TB >= 1) { const uint32_t pv = pslot_lane_xor<1>(v), pj = pslot_lane_xor<1>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[0]); v = take ? cand : v; j = take ? pj : j; }
		if (TB >= 2) { const uint32_t pv = pslot_lane_xor<2>(v), pj = pslot_lane_xor<2>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[TB >= 2 ? 1 : 0]); v = take ? cand : v; j = take ? pj : j; }
		if (TB >= 3) { const uint32_t pv = pslot_lane_xor<4>(v), pj = pslot_lane_xor<4>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[TB >= 3 ? 2 : 0]); v = take ? cand : v; j = take ? pj : j; }
		if (TB >= 4) { const uint32_t pv = pslot_lane_xor<8>(v), pj = pslot_lane_xor<8>(j); const uint32_t cand = pslot_sat_add(pv, rc); const bool take = cand < pslot_sat_add(v, tbit[TB >= 4 ? 3 : 0]); v = take ? cand : v; j = take ? pj : j; }
		D = pslot_sat_add(v, cst);
		uint32_t byte = j;
		const uint32_t n_end = ctrl & 7u;
		if (n_end) {
			pslotx_ending<4, TB>(D, byte, par, ctrl >> 3, xbytes, lane4, tid4);
			if (n_end > 1u) {   // several reads ending in one column are rare; all four fields are in the control word
				pslotx_ending<5, TB>(D, byte, par, ctrl >> 7, xbytes, lane4, tid4);
				if (n_end > 2u) {
					pslotx_ending<6, TB>(D, byte, par, ctrl >> 11, xbytes, lane4, tid4);
					if (n_end > 3u) pslotx_ending<7, TB>(D, byte, par, ctrl >> 15, xbytes, lane4, tid4);
				}
			}
		}
		return byte;
	};
	const uint32_t xstride = threads * 16u;
	uint32_t xaddr = lds0 + 2u * xbytes + (tid << 4);   // LDS byte address of the lane's cost line of trip 0 (behind the two exchange buffers)
	typedef __attribute__((address_space(3))) const u32x4c* lds_line;
	u32x4c xA = *(lds_line)(size_t)xaddr, xB;
	for (uint32_t pairs = (ncols + 7u) >> 3; pairs; --pairs) {
		asm volatile("" ::"s"(rcA[0]), "s"(cwA[0]), "v"(xA[0]));
		rcB = *(cptr4)(unsigned long long)(rc_tab + 4u);
		cwB = *(cptr4)(unsigned long long)(cw_tab + 4u);
		xB = *(lds_line)(size_t)(xaddr + xstride);
		uint32_t recacc = column(xA[0], rcA[0], cwA[0]);
		recacc |= column(xA[1], rcA[1], cwA[1]) << 8;
		recacc |= column(xA[2], rcA[2], cwA[2]) << 16;
		recacc |= column(xA[3], rcA[3], cwA[3]) << 24;
		rec[0] = recacc;   // fire and forget (a trip behind the run's last writes a word of the next run's record: never -- the planner's rec_words is whole trips,
		                   // and a pad trip only exists in the second half below)
		asm volatile("" ::"s"(rcB[0]), "s"(cwB[0]), "v"(xB[0]));
		rc_tab += 8u;
		cw_tab += 8u;
		xaddr += 2u * xstride;
		rcA = *(cptr4)(unsigned long long)rc_tab;
		cwA = *(cptr4)(unsigned long long)cw_tab;
		xA = *(lds_line)(size_t)xaddr;
		recacc = column(xB[0], rcB[0], cwB[0]);
		recacc |= column(xB[1], rcB[1], cwB[1]) << 8;
		recacc |= column(xB[2], rcB[2], cwB[2]) << 16;
		recacc |= column(xB[3], rcB[3], cwB[3]) << 24;
		if (pairs > 1u || ((ncols + 3u) >> 2 & 1u) == 0u) rec[threads] = recacc;   // (the second trip of the last pair may lie behind the run)
		rec += 2u * (size_t)threads;
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
#pragma unroll
			for (int m = 1; m < 64; m <<= 1) {
				const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best_key, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best_key >> 32), m);
				best_key = min(best_key, ((unsigned long long)hi << 32) | lo);
			}
			if (lane == 0) P.spec_keys[(size_t)(run.spec_id - 1u) * P.spec_stride + w * (threads >> 6) + wave] = best_key;
		}
	}
}
inline size_t pedslotx_lds_bytes(uint32_t threads, uint32_t xc) { return (size_t)threads * 8 + (size_t)(xc / 4 + 3) * threads * 16; }

template <int TB, int NF, int XC, bool SPEC>
__global__ __launch_bounds__(512) void pedslot_runx(DevProblem P, SlotRun run, PedSlotExtra ex, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur) {
	touch_kernel_arguments<sizeof(DevProblem) + sizeof(SlotRun) + sizeof(PedSlotExtra) + 16>();
	pedslot_runx_body<TB, NF, XC, SPEC>(P, run, ex, prev, cur, blockIdx.x);
}
#endif

template <int TB, int NF, bool SPEC, bool PACKED>
__global__ __launch_bounds__(512) void pedslot_run(DevProblem P, SlotRun run, PedSlotExtra ex, const uint32_t* __restrict__ prev,
                                                   uint32_t* __restrict__ cur) {
	touch_kernel_arguments<sizeof(DevProblem) + sizeof(SlotRun) + sizeof(PedSlotExtra) + 16>();
	pedslot_run_body<TB, NF, SPEC, PACKED>(P, run, ex, prev, cur, blockIdx.x);
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
	if (run.yflags & 16u) {   // the min-plus step on packed keys (one branch per launch, wave-uniform)
		if (run.spec_id) pedslot_run_body<TB, NF, true, true>(P, run, e.ex, e.prev, e.cur, who.w);
		else pedslot_run_body<TB, NF, false, true>(P, run, e.ex, e.prev, e.cur, who.w);
	} else {
		if (run.spec_id) pedslot_run_body<TB, NF, true, false>(P, run, e.ex, e.prev, e.cur, who.w);
		else pedslot_run_body<TB, NF, false, false>(P, run, e.ex, e.prev, e.cur, who.w);
	}
	slot_warm_done(warm);
}
