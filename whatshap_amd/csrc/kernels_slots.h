// kernels_slots.h -- slot_run: the register-resident run kernel of a single individual (slots.h has the design).
// Included by dp_device.hip inside namespace whamd { namespace { ... } }: not a stand-alone header.
//
// One launch = one run of consecutive columns.  Workgroup w, wave v, lane l and register r of a thread hold the cell
// with physical index P = (w << L) | (v << (6 + LR)) | (l << LR) | r for the whole run; a column adds its closed-form
// cost to every cell, an ending read is minimised out where it sits (registers / cross-lane move / LDS exchange
// between two waves), a starting read just begins to contribute its delta.  Per-column constants are wave-uniform: the
// hot lines of the run's columns are staged in LDS by the prologue and read back three columns ahead (wave-uniform words in
// VECTOR registers); the workgroup- and wave-dependent part of S and the lane sums come from tables built once per table
// (slot_tables); what steers control flow -- does a read end in this column, in which slot -- is one control byte per column,
// sixteen words in the lanes of one register, fetched per trip of four columns with a v_readlane.
//
// Restates compute_column of the reference (src/pedigreedptable.cpp:177-335) for T = 1: cost per cell (:262-283 with
// PedigreeColumnCostComputer::get_cost), strict-'<' projection with Gray-code visiting order (:306-327) as the tie
// rule of slots.h.

// min(Cp + S, Cm - S, Cc) with A = Cp + S and K = Cp + Cm (mod 2^32; absent terms are RES_ABSENT and never the minimum)
__device__ __forceinline__ uint32_t slot_cost(uint32_t A, uint32_t K, uint32_t Cc) { return min(min(A, K - A), Cc); }

// One ending read whose slot is reg slot J: the two cells of a pair live in the same thread.
template <int LR, int J>
__device__ __forceinline__ uint32_t slot_end_reg(uint32_t (&D)[1 << LR], uint32_t qthr, uint32_t qmask) {
	constexpr int R = 1 << LR;
	uint32_t takes = 0;
#pragma unroll
	for (int r0 = 0; r0 < R; ++r0) {
		if (r0 & (1 << J)) continue;
		const int r1 = r0 | (1 << J);
		const uint32_t q0 = qthr ^ ((qmask >> r0) & 1u), q1 = qthr ^ ((qmask >> r1) & 1u);
		const uint32_t a = D[r0], b = D[r1];
		takes |= (b < a + q0) ? (1u << r0) : 0u;   // the pair's decision: side 1 wins if smaller, or equal and favoured
		takes |= (a < b + q1) ? (1u << r1) : 0u;   // the decision of the pair's mirror image
		D[r0] = D[r1] = min(a, b);
	}
	return takes;
}

// The partner cells are held by `other` (same register index r): lane slot (cross-lane move) or wave slot (LDS).
template <int LR>
__device__ __forceinline__ uint32_t slot_end_partner(uint32_t (&D)[1 << LR], const uint32_t (&other)[1 << LR], uint32_t qthr, uint32_t qmask) {
	constexpr int R = 1 << LR;
	uint32_t takes = 0;
#pragma unroll
	for (int r = 0; r < R; ++r) {
		const uint32_t q = qthr ^ ((qmask >> r) & 1u);
		takes |= (other[r] < D[r] + q) ? (1u << r) : 0u;   // side-0 cell: decision of the pair; side-1 cell: of its mirror image
		D[r] = min(D[r], other[r]);
	}
	return takes;
}

// Wave-uniform data is read through the scalar cache: loads from the constant address space become s_load_dwordx8
// (kernel arguments are the same kind of memory).  A generic pointer converts bit for bit.
typedef uint32_t slot_u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t slot_u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t slot_u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) slot_u32x8* slot_cptr8;
typedef const __attribute__((address_space(4))) slot_u32x16* slot_cptr16;
typedef const __attribute__((address_space(4))) slot_u32x2* slot_cptr2;
template <class T>
__device__ __forceinline__ slot_cptr8 slot_scalar_ptr(const T* p) { return (slot_cptr8)(unsigned long long)p; }
// byte s of a packed position table held in SGPRs (static s)
__device__ __forceinline__ uint32_t slot_pos_dev(const uint32_t (&w)[8], int s) { return (w[s >> 2] >> ((s & 3) * 8)) & 31u; }

// ---- Y form (slots.h, slot_plan.cpp): a run of columns that all have both orientation terms and no constant one keeps
// Y = B_c - 2 D per cell (B_c: the same for every cell of column c) instead of D.  2 min(A, K - A) = K - |2A - K|, so a cell-column is
// ONE instruction, Y += |X0 - Kr[r]| with X0 = 2 A(cell 0 of the thread) + bias from the (doubled) tables and Kr[r] = K - 2 dreg(r) + bias
// from the hot line -- v_sad_u32, which issues at the rate of v_min3_u32 alone (measured, scripts/micro/op_rates.hip: add / sub / xor
// ~1.0 ns per wave-instruction and SIMD at saturation, min / max / min3 / sad / add3 / dpp ~1.75 ns).  Minima over ending reads become
// maxima; "other < mine + q" becomes "Y_mine < Y_other + q": the BORROW of Y_mine - Y_other - q (v_subb_co_u32 with q as the carry-in
// lane mask), shifted into the record byte by v_addc_co_u32 (takes + takes + borrow): three instructions per cell and ending read.
__device__ __forceinline__ uint32_t slot_y_sad(uint32_t x, uint32_t k, uint32_t acc) {
	uint32_t o;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(o) : "v"(x), "v"(k), "v"(acc));
	return o;
}
// (the second operand wave-uniform, in an SGPR: X runs)
__device__ __forceinline__ uint32_t slot_y_sad_s(uint32_t x, uint32_t k, uint32_t acc) {
	uint32_t o;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(o) : "v"(x), "s"(k), "v"(acc));
	return o;
}
__device__ __forceinline__ unsigned long long slot_y_borrow(uint32_t mine, uint32_t other, unsigned long long q) {
	uint32_t d;
	unsigned long long bo;
	asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d), "=s"(bo) : "v"(mine), "v"(other), "s"(q));
	return bo;
}
__device__ __forceinline__ uint32_t slot_y_shift_in(uint32_t takes, unsigned long long bit) {
	uint32_t o;
	unsigned long long co;
	asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(o), "=s"(co) : "v"(takes), "s"(bit));
	return o;
}

// The entering cells of a run (raw loads only: nothing consumes them before the prologue has everything else in flight).  `flip`: the thread's
// cells come from the mirrored group of a halved run, in reverse order.
template <int LR, bool DBG>
__device__ __forceinline__ void slot_enter_cells(const DevProblem& P, const SlotRun& run, const uint32_t* __restrict__ prev, const uint32_t Pthr,
                                                 uint32_t (&Draw)[1 << LR], bool& flip) {
	constexpr int R = 1 << LR;
	flip = false;
	if (run.has_prev && !(DBG && (P.dbg_flags & 64u))) {
		const uint32_t occ = run.in_occ;
		if (run.in_identity && (occ & (uint32_t)(R - 1)) == (uint32_t)(R - 1) && (!run.in_half || run.in_mirror_pos >= (uint32_t)LR)) {
			// the previous run stored in this run's physical order: R contiguous entries per thread; after a halved run
			// the entries whose mirror bit is set come from the complement index, i.e. the mirrored group in reverse
			uint32_t base = Pthr & occ;
			flip = run.in_half && ((base >> run.in_mirror_pos) & 1u);
			if (flip) base = (base ^ run.in_fullmask) & ~(uint32_t)(R - 1);
#pragma unroll
			for (int q = 0; q < R / 4; ++q) {
				const uint4 t = *reinterpret_cast<const uint4*>(prev + base + 4 * q);
				Draw[4 * q] = t.x; Draw[4 * q + 1] = t.y; Draw[4 * q + 2] = t.z; Draw[4 * q + 3] = t.w;
			}
			if (R == 2) {   // two cells per thread: one 8-byte load
				const uint2 t = *reinterpret_cast<const uint2*>(prev + base);
				Draw[0] = t.x; Draw[R - 1] = t.y;
			}
		} else {
			// any layout (the writer's order inside this workgroup's block; logical order after a per-column step): index bit by
			// bit, tables in SGPRs, static slot indices
			uint32_t pos[SLOT_MAXSLOTS];
#pragma unroll
			for (int s = 0; s < SLOT_MAXSLOTS; ++s) pos[s] = run.in_identity ? (uint32_t)s : slot_pos_dev(run.in_pos, s);
			uint32_t base = 0;
#pragma unroll
			for (int s = LR; s < SLOT_MAXSLOTS; ++s) base |= ((Pthr & occ) >> s & 1u) << pos[s];
#pragma unroll
			for (int r = 0; r < R; ++r) {
				uint32_t idx = base;
#pragma unroll
				for (int s = 0; s < LR; ++s)
					if ((r >> s) & 1) idx |= ((occ >> s) & 1u) << pos[s];
				if (run.in_half && ((idx >> run.in_mirror_pos) & 1u)) idx ^= run.in_fullmask;
				Draw[r] = prev[idx];
			}
		}
	} else {
#pragma unroll
		for (int r = 0; r < R; ++r) Draw[r] = 0;
	}
}

// ---- exit of a run: scatter into the next step's order (cells whose free-slot bits are zero hold the representatives), the seed of the
// speculative backtrace (SPEC), Y form back to D form where the next step is not a Y-form run.
template <int LR, bool DBG, bool SPEC, bool YF>
__device__ __forceinline__ void slot_exit_cells(const DevProblem& P, const SlotRun& run, uint32_t* __restrict__ cur, uint32_t (&D)[1 << LR], const uint32_t w,
                                                const uint32_t tid, const uint32_t lane, const uint32_t wave, const uint32_t L, const uint32_t lthr, const uint32_t Pthr,
                                                const uint32_t threads) {
	constexpr int R = 1 << LR;
	const bool y_out = YF && (run.yflags & 4u);   // the next step is a Y-form run too: the column stays as it is
	if (YF && !y_out) {
#pragma unroll
		for (int r = 0; r < R; ++r) D[r] = (run.base_out - D[r]) >> 1;   // D = (B - Y) / 2, exactly
	}
	const uint32_t key_flip = y_out ? 0xFFFFFFFFu : 0u;   // seeds of the speculative backtrace order by D: the largest Y is the smallest D
	{
		const uint32_t occ = run.out_occ;
		const uint32_t localmask = (1u << L) - 1u;
		const bool thread_writes = ((lthr & ~(uint32_t)(R - 1)) & ~occ & localmask) == 0u;
		uint32_t pos[SLOT_MAXSLOTS];
#pragma unroll
		for (int s = 0; s < SLOT_MAXSLOTS; ++s) pos[s] = slot_pos_dev(run.out_pos, s);
		uint32_t base = 0;
#pragma unroll
		for (int s = LR; s < SLOT_MAXSLOTS; ++s) base |= ((Pthr & occ) >> s & 1u) << pos[s];
		const uint32_t mirror_x = run.mirror_out ? run.out_fullmask : 0u;
		unsigned long long best_key = ~0ull;   // (value, exit index) of the smallest cell this thread stores (run.spec_id)
		if (R == 2 && (occ & 1u) && pos[0] == 0u) {
			// two cells per thread, the read of the reg slot is the lowest bit of the exit index: one 8-byte store
			if (thread_writes && !(DBG && (P.dbg_flags & 1u))) {
				*reinterpret_cast<uint2*>(cur + base) = make_uint2(D[0], D[R - 1]);
				if (SPEC) best_key = min(min(best_key, ((unsigned long long)(D[0] ^ key_flip) << 32) | base), ((unsigned long long)(D[R - 1] ^ key_flip) << 32) | (base + 1u));
				if (run.mirror_out) *reinterpret_cast<uint2*>(cur + ((base ^ mirror_x) & ~1u)) = make_uint2(D[R - 1], D[0]);
			}
		} else if (R >= 4 && (occ & 3u) == 3u && pos[0] == 0u && pos[1] == 1u) {
			// the reads of reg slots 0 and 1 are the two lowest bits of the exit index (the planner arranges that for reads
			// that stay local in the next run): 4 cells = one 16-byte store
#pragma unroll
			for (int r4 = 0; r4 < R; r4 += 4) {
				bool writes = thread_writes;
				uint32_t x = 0;
#pragma unroll
				for (int s = 2; s < LR; ++s) {
					if ((r4 >> s) & 1) {
						x |= 1u << pos[s];
						writes = writes && ((occ >> s) & 1u);
					}
				}
				if (writes && !(DBG && (P.dbg_flags & 1u))) {
					const uint32_t idx = base | x;
					*reinterpret_cast<uint4*>(cur + idx) = make_uint4(D[r4], D[r4 + 1], D[r4 + 2], D[r4 + 3]);
					if (SPEC) {
#pragma unroll
						for (int j = 0; j < 4; ++j) best_key = min(best_key, ((unsigned long long)(D[r4 + j] ^ key_flip) << 32) | (idx + j));
					}
					if (run.mirror_out)   // the complement of a group of 4 is a group of 4 in reverse order
						*reinterpret_cast<uint4*>(cur + ((idx ^ mirror_x) & ~3u)) = make_uint4(D[r4 + 3], D[r4 + 2], D[r4 + 1], D[r4]);
				}
			}
		} else {
#pragma unroll
			for (int r = 0; r < R; ++r) {
				bool writes = thread_writes;
				uint32_t x = 0;
#pragma unroll
				for (int s = 0; s < LR; ++s) {
					if ((r >> s) & 1) {
						x |= 1u << pos[s];
						writes = writes && ((occ >> s) & 1u);
					}
				}
				if (writes && !(DBG && (P.dbg_flags & 1u))) {
					const uint32_t idx = base | x;
					cur[idx] = D[r];
					if (run.mirror_out) cur[idx ^ mirror_x] = D[r];
					if (SPEC) best_key = min(best_key, ((unsigned long long)(D[r] ^ key_flip) << 32) | idx);
				}
			}
		}
		if (SPEC && run.spec_id) {
			// Seed of the speculative backtrace (kernels_backtrace.h): the smallest entry of the exit column.  Any entry would
			// keep the result exact (the walk from the seed is verified against the true path); the minimum is what the true
			// path almost always runs through.  One candidate per wave.
#pragma unroll
			for (int m = 1; m < 64; m <<= 1) {
				const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best_key, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best_key >> 32), m);
				best_key = min(best_key, ((unsigned long long)hi << 32) | lo);
			}
			// one plain store per wave (2048 atomics on one word would take ~25 us); the backtrace reduces the candidates
			if (lane == 0) P.spec_keys[(size_t)(run.spec_id - 1u) * P.spec_stride + w * (threads >> 6) + wave] = best_key;
		}
	}
}

template <int LR, bool DBG, bool SPEC, bool YF = false>
__device__ __forceinline__ void slot_run_body(const DevProblem& P, const SlotRun& run, const uint32_t* __restrict__ prev,
                                              uint32_t* __restrict__ cur, const uint32_t w, uint32_t* score_out) {
	constexpr int R = 1 << LR;
	static_assert(!YF || LR == 2 || LR == 3, "Y-form rows hold Kr[0 .. 2^LR): four or eight cells per thread");
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // wave-slot exchange: 2 x [threads][R]
	const unsigned long long t_start = (DBG && P.dbg) ? __builtin_readcyclecounter() : 0ull;
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t wave = uni(tid >> 6);
	const uint32_t L = run.L;
	const uint32_t lthr = tid << LR;               // local index of this thread's cell 0
	const uint32_t Pthr = (w << L) | lthr;         // its physical index
	const SlotRow* __restrict__ rows = P.slot_rows + run.row_off;
	const uint32_t ncols = run.ncols;

	// ---- prologue: ONE batch of global loads, issued before anything waits (vector loads return in order, so the first
	// consumers below wait only for what was issued first).
	// (1) lane c of every wave fetches A of column c = Cp + (deltas of the set grid slots: table G of this workgroup) + (deltas of
	//     the set wave slots: table W of this wave) -- slot_tables built both at create time; the cold part of SlotRow (80 bytes per
	//     lane, an 18-slot loop) is no longer touched by a run
	uint32_t a_g, a_w;
	{
		const uint32_t cl = lane < ncols ? lane : 0u;
		a_g = a_w = 0;
		const uint32_t ncp = (run.yflags & 8u) ? (ncols + 7u) & ~7u : ncols;   // (the tables of a run that may take the X kernel have padded rows)
		if (!(DBG && (P.dbg_flags & 128u))) {   // (WHAMD_SLOT_SKIP 32 / 64 / 128: prologue loads switched off -- lane sums / entering cells / A, hot lines)
			a_g = P.slot_tab[run.tab_g + w * ncp + cl];
			a_w = P.slot_tab[run.tab_w + wave * ncp + cl];
		}
	}
	//     ... and the lane part of S(column, lane), table SL: the same for every workgroup, 16 bytes per thread
	const uint4* __restrict__ sl_src = reinterpret_cast<const uint4*>(P.slot_tab + run.tab_sl);
	uint4 sl_piece = make_uint4(0, 0, 0, 0);
	if (tid < ncols * 16u && !(DBG && (P.dbg_flags & 32u))) sl_piece = sl_src[tid];
	// (2) one 16-byte piece of the hot lines per thread (they go to LDS below)
	uint4 hot_piece = make_uint4(0, 0, 0, 0);
	if (tid < ncols * 4u && !(DBG && (P.dbg_flags & 128u))) hot_piece = reinterpret_cast<const uint4*>(rows + (tid >> 2))[tid & 3u];
	// (3) the entering cells: raw loads only (nothing consumes them before the hot lines and the lane sums are in LDS)
	uint32_t Draw[R];
	bool flip;
	slot_enter_cells<LR, DBG>(P, run, prev, Pthr, Draw, flip);
	// The hot lines of the run's columns go to LDS.  The column loop reads them back with uniform-address LDS reads one column
	// ahead: LDS returns in order (lgkmcnt), so the read of the next column stays in flight while this one is evaluated --
	// scalar loads cannot do that (they return out of order: every wait drains them all, and even a scalar-cache hit costs
	// ~300 cycles), and v_readlane broadcasts cost ~35 cycles each.
	uint32_t* hot_lds = smem + 2u * run.threads * R;
	if (tid < ncols * 4u) reinterpret_cast<uint4*>(hot_lds)[tid] = hot_piece;
	for (uint32_t i = tid + run.threads; i < ncols * 4u; i += run.threads)   // narrow workgroups, long runs
		reinterpret_cast<uint4*>(hot_lds)[i] = reinterpret_cast<const uint4*>(rows + (i >> 2))[i & 3u];
	const uint32_t Avec = a_g + a_w;
	// ... and so does A of every column, one 64-entry row per wave: a VALU -> SGPR transfer (v_readlane, v_readfirstlane)
	// costs ~35 cycles of issue, so nothing on the column chain goes that way.  What steers control flow (does a read end in
	// this column, in which slot) comes from the run's control bytes, loaded once into SGPRs.
	const uint32_t hot_words = (ncols + 8u) * 16u;   // (the LDS areas follow the run's own length: slot_run_lds_bytes)
	uint32_t* a_lds = hot_lds + hot_words + wave * 64u;
	a_lds[lane] = Avec;
	uint32_t* sl_lds = hot_lds + hot_words + 8 * 64;   // [column][lane]: lane part of S, the same for every wave
	if (tid < ncols * 16u) reinterpret_cast<uint4*>(sl_lds)[tid] = sl_piece;
	for (uint32_t i = tid + run.threads; i < ncols * 16u; i += run.threads) reinterpret_cast<uint4*>(sl_lds)[i] = sl_src[i];   // narrow workgroups, long runs
	// the run's 16 control words (one byte per column), one per lane: the word of a trip is fetched with ONE v_readlane (a queue of 16
	// SGPRs rotated with scalar moves was 17 instructions per trip of four columns -- an eighth of a plain column's instructions)
	const uint32_t ctrl_v = P.slot_ctrl[run.ctrl_off + (lane & 31u)];   // (16 bits per column: SLOT_CTRL_WORDS = 32)
	uint8_t* __restrict__ rec = P.bt + (((unsigned long long)run.rec_hi << 32) | run.rec_lo) + (size_t)w * run.n_ends * run.threads + tid;
	const uint32_t threads = run.threads;
	const uint32_t xwords = threads * R;   // one exchange buffer
	uint32_t xsel = 0;
	const unsigned long long t_issued = (DBG && P.dbg) ? __builtin_readcyclecounter() : 0ull;
	__syncthreads();
	uint32_t D[R];   // (Y form: Y = B - 2 D)
#pragma unroll
	for (int r = 0; r < R; ++r) D[r] = flip ? Draw[R - 1 - r] : Draw[r];
	if (YF && !(run.yflags & 2u)) {   // the entering column is in D form (a per-column step, a run of the other kind, a fresh component)
#pragma unroll
		for (int r = 0; r < R; ++r) D[r] = run.base_in - 2u * D[r];
	}
	if (DBG && P.dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	const unsigned long long t_loaded = (DBG && P.dbg) ? __builtin_readcyclecounter() + (D[0] & 0u) : 0ull;

	// What a column needs from LDS, requested one column ahead: {K, Cc, dreg0, dreg1} (+ dreg2), {info0, M0} of the first ending
	// read (wave-uniform words in VECTOR registers: operands of the cell arithmetic as they are) and the thread's own A.
	struct HotLine { uint4 a; uint4 b; uint32_t d2; uint2 e; uint32_t A; };   // (b: Kr[4..7] of a Y-form run with eight cells per thread)
	// The lines are requested in column order, so three running word offsets (hot line, A of the wave, lane sum) advance by a constant
	// per request: three adds instead of rebuilding each address from the column number (5 vector + 3 scalar instructions of the ~34 a
	// plain column took).  They start from an opaque move: the compiler must not learn that the hot-line loads are wave-uniform, or it
	// selects scalar instructions for what is derived from them and pays a v_readfirstlane for every operand.
	// (three LDS pointers that advance once per trip of four columns; inside a trip every request is pointer + a constant that folds
	//  into the ds_read offset field -- as running offsets advanced per request the compiler rebuilt all three addresses for every column)
	uint32_t hot_at = 0, a_at = 0, sl_at = lane;
	asm volatile("" : "+v"(hot_at), "+v"(a_at), "+v"(sl_at));
	const uint32_t* hot_p = hot_lds + hot_at;
	const uint32_t* a_p = a_lds + a_at;
	const uint32_t* sl_p = sl_lds + sl_at;
	auto load_hot = [&](const uint32_t k) -> HotLine {   // line k of the current trip (k = 0 .. 6: a line is requested three columns ahead)
		HotLine h;
		h.a = *reinterpret_cast<const uint4*>(hot_p + 16u * k);
		h.d2 = (LR > 2 && !YF) ? hot_p[16u * k + 4u] : 0u;
		if (YF && LR > 2) h.b = *reinterpret_cast<const uint4*>(hot_p + 16u * k + 4u);
		h.e = *reinterpret_cast<const uint2*>(hot_p + 16u * k + 12u);
		h.A = a_p[k] + sl_p[64u * k];
		return h;
	};
	// One column for the calling thread's cells.  The hot words are wave-uniform values in VECTOR registers: operands of the
	// cell arithmetic as they are; only what steers control flow (n_end, the ending read's slot) becomes scalar.
	auto column = [&](const HotLine& h, const uint32_t ci, const uint32_t ctrl) {
		if (YF) {
			const uint32_t kr[8] = {h.a.x, h.a.y, h.a.z, h.a.w, h.b.x, h.b.y, h.b.z, h.b.w};
#pragma unroll
			for (int r = 0; r < R; ++r) D[r] = slot_y_sad(h.A, kr[r & 7], D[r]);
		} else {
			const uint32_t K = h.a.x, Cc = h.a.y;
			const uint32_t dr[SLOT_LR + 1] = {h.a.z, h.a.w, h.d2, 0u};
			const uint32_t A = h.A;
			uint32_t Ar[R];
			Ar[0] = A;
#pragma unroll
			for (int r = 1; r < R; ++r) Ar[r] = Ar[r & (r - 1)] + dr[__builtin_ctz(r)];   // clear the lowest set bit: one add per cell
#pragma unroll
			for (int r = 0; r < R; ++r) D[r] += slot_cost(Ar[r], K, Cc);
		}
		const uint32_t n_end = (DBG && (P.dbg_flags & 8u)) ? 0u : (ctrl & 3u);
		// Y form: one ending read.  `slot` and `qmask` are scalars, M a wave-uniform value (vector or scalar register).
		auto ending_y = [&](const uint32_t M, const uint32_t slot, const uint32_t qmask) {
			const uint32_t par = (uint32_t)__popc(Pthr & M) & 1u;
			const unsigned long long QT = __builtin_amdgcn_uicmp(par, 0u, 33);   // lanes whose thread parity is odd
			const unsigned long long QN = ~QT;
			unsigned long long Q[R];
#pragma unroll
			for (int r = 0; r < R; ++r) Q[r] = ((qmask >> r) & 1u) ? QN : QT;   // scalar unit: s_bitcmp1 + s_cselect_b64 per cell
			uint32_t other[R];
			if (slot < (uint32_t)LR) {
				if (slot == 0) {
#pragma unroll
					for (int r = 0; r < R; ++r) other[r] = D[r ^ 1];
				} else if (LR < 3 || slot == 1) {
#pragma unroll
					for (int r = 0; r < R; ++r) other[r] = D[r ^ (R > 2 ? 2 : 1)];
				} else {
#pragma unroll
					for (int r = 0; r < R; ++r) other[r] = D[r ^ (R > 4 ? 4 : 1)];
				}
			} else if (slot < (uint32_t)(LR + SLOT_LANE)) {
				const int src = (int)((lane ^ (1u << (slot - LR))) << 2);
#pragma unroll
				for (int r = 0; r < R; ++r) other[r] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)D[r]);
			} else {
				uint32_t* xb = smem + xsel * xwords;
#pragma unroll
				for (int q4 = 0; q4 < R / 4; ++q4) reinterpret_cast<uint4*>(xb + tid * R)[q4] = make_uint4(D[4 * q4], D[4 * q4 + 1], D[4 * q4 + 2], D[4 * q4 + 3]);
				__syncthreads();
				const uint32_t ptid = tid ^ (64u << (slot - LR - SLOT_LANE));
#pragma unroll
				for (int q4 = 0; q4 < R / 4; ++q4) {
					const uint4 t = reinterpret_cast<const uint4*>(xb + ptid * R)[q4];
					other[4 * q4] = t.x; other[4 * q4 + 1] = t.y; other[4 * q4 + 2] = t.z; other[4 * q4 + 3] = t.w;
				}
			}
			xsel ^= (uint32_t)(slot >= (uint32_t)(LR + SLOT_LANE));   // (outside the branches: a scalar update, no merge of branch values)
			uint32_t takes = 0;
#pragma unroll
			for (int r = R - 1; r >= 0; --r) takes = slot_y_shift_in(takes, slot_y_borrow(D[r], other[r], Q[r]));   // bit r: this cell's decision
#pragma unroll
			for (int r = 0; r < R; ++r) D[r] = max(D[r], other[r]);
			if (!(DBG && (P.dbg_flags & 2u))) *rec = (uint8_t)takes;
			rec += threads;
		};
		// one ending read: `slot` is a scalar (control flow), info / M are wave-uniform vector values
		auto ending = [&](const uint32_t info, const uint32_t M, const uint32_t slot) {
			const uint32_t qmask = (info >> 8) & 0xFFFFu;   // (lane / wave slots: the side term of the parity is folded into M by the planner)
			uint32_t qthr = (uint32_t)__popc(Pthr & M) & 1u;
			uint32_t takes;
			if (slot < (uint32_t)LR) {
				if (slot == 0) takes = slot_end_reg<LR, 0>(D, qthr, qmask);
				else if (LR > 1 && slot == 1) takes = slot_end_reg<LR, (LR > 1 ? 1 : 0)>(D, qthr, qmask);
				else if (LR > 2 && slot == 2) takes = slot_end_reg<LR, (LR > 2 ? 2 : 0)>(D, qthr, qmask);
				else takes = slot_end_reg<LR, (LR > 3 ? 3 : 0)>(D, qthr, qmask);
			} else if (slot < (uint32_t)(LR + SLOT_LANE)) {
				const int src = (int)((lane ^ (1u << (slot - LR))) << 2);
				uint32_t other[R];
#pragma unroll
				for (int r = 0; r < R; ++r) other[r] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)D[r]);
				takes = slot_end_partner<LR>(D, other, qthr, qmask);
			} else {
				// the partner cells are another wave's registers: exchange through LDS (two buffers: the barrier of the next
				// exchange also protects this one's reads)
				uint32_t* xb = smem + xsel * xwords;
				uint4* mine = reinterpret_cast<uint4*>(xb + tid * R);
#pragma unroll
				for (int q4 = 0; q4 < R / 4; ++q4) mine[q4] = make_uint4(D[4 * q4], D[4 * q4 + 1], D[4 * q4 + 2], D[4 * q4 + 3]);
				if (R == 2) *reinterpret_cast<uint2*>(mine) = make_uint2(D[0], D[R - 1]);
				__syncthreads();
				const uint32_t ptid = tid ^ (64u << (slot - LR - SLOT_LANE));
				const uint4* theirs = reinterpret_cast<const uint4*>(xb + ptid * R);
				uint32_t other[R];
#pragma unroll
				for (int q4 = 0; q4 < R / 4; ++q4) {
					const uint4 t = theirs[q4];
					other[4 * q4] = t.x; other[4 * q4 + 1] = t.y; other[4 * q4 + 2] = t.z; other[4 * q4 + 3] = t.w;
				}
				if (R == 2) {
					const uint2 t = *reinterpret_cast<const uint2*>(theirs);
					other[0] = t.x; other[R - 1] = t.y;
				}
				takes = slot_end_partner<LR>(D, other, qthr, qmask);
				xsel ^= 1u;
			}
			if (!(DBG && (P.dbg_flags & 2u))) *rec = (uint8_t)takes;
			rec += threads;
		};
		if (DBG && P.dbg && w == 0 && tid == 0 && ci < 32u) P.dbg[(size_t)run.pad * 48 + 8 + ci] = __builtin_readcyclecounter() - t_loaded;
		if (n_end) {
			if (YF) ending_y(h.e.y, (ctrl >> 2) & 31u, (ctrl >> 7) & 255u);
			else ending(h.e.x, h.e.y, (ctrl >> 2) & 31u);
			if (n_end > 1u) {   // several reads ending in one column (rare): their slots come out of the hot line -- a VALU -> SGPR
			                    // transfer, kept out of the common path (volatile: must not be hoisted above this branch)
				uint32_t off1 = ci * 16u + 14u;
				asm volatile("" : "+v"(off1));
				const uint2 e1 = *reinterpret_cast<const uint2*>(hot_lds + off1);
				uint32_t s1;
				asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1" : "=s"(s1) : "v"(e1.x));
				if (YF) ending_y(e1.y, s1 & 255u, (s1 >> 8) & 255u);
				else ending(e1.x, e1.y, s1 & 255u);
				if (n_end > 2u) {   // the third and later lie in the row's second line; a control byte of 3 says "three or more"
					const uint32_t total = *(const __attribute__((address_space(4))) uint32_t*)((unsigned long long)(rows + ci) + 44);
					for (uint32_t e = 2; e < total; ++e) {
						const slot_u32x2 ex = *(slot_cptr2)((unsigned long long)(rows + ci) + 48 + 8 * e);
						if (YF) ending_y(ex[1], ex[0] & 255u, (ex[0] >> 8) & 255u);
						else ending(ex[0], ex[1], ex[0] & 255u);
					}
				}
			}
		}
	};
	{
		// two columns per trip: each column's hot line is requested while the previous column is evaluated, without register
		// copies; the control byte of column c is byte c of the 16 control words (static word index: four columns per word)
		const uint32_t nc = (DBG && (P.dbg_flags & 4u)) ? 1u : ncols;
		// four columns per trip (one control word), four line buffers: every line is requested THREE columns ahead (LDS returns
		// in order, the waits count down) and no register is copied
		HotLine h0 = load_hot(0u), h1 = load_hot(1u), h2 = load_hot(2u), h3;
		for (uint32_t ci = 0; ci < nc; ci += 4u) {
			// control words of columns ci, ci + 1 and ci + 2, ci + 3 (16 bits per column)
			const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)ctrl_v, (int)(ci >> 1));
			h3 = load_hot(3u);             // (lines beyond the run may be read: the LDS areas have room, the values are not used)
			column(h0, ci, cw & 0xFFFFu);
			if (ci + 1u >= nc) break;
			h0 = load_hot(4u);
			column(h1, ci + 1u, cw >> 16);
			if (ci + 2u >= nc) break;
			const uint32_t cw2 = (uint32_t)__builtin_amdgcn_readlane((int)ctrl_v, (int)((ci >> 1) + 1u));
			h1 = load_hot(5u);
			column(h2, ci + 2u, cw2 & 0xFFFFu);
			if (ci + 3u >= nc) break;
			h2 = load_hot(6u);
			column(h3, ci + 3u, cw2 >> 16);
			hot_p += 64u;
			a_p += 4u;
			sl_p += 256u;
		}
	}

	const unsigned long long t_loop = (DBG && P.dbg) ? __builtin_readcyclecounter() + (D[0] & 0u) : 0ull;
	slot_exit_cells<LR, DBG, SPEC, YF>(P, run, cur, D, w, tid, lane, wave, L, lthr, Pthr, threads);
	if (score_out && w == 0 && tid == 0) *score_out = D[0];
	if (DBG && P.dbg && w == 0 && tid == 0) {
		unsigned long long* d = P.dbg + (size_t)run.pad * 48;
		d[0] = t_issued - t_start; d[1] = t_loaded - t_start; d[2] = t_loop - t_loaded; d[3] = __builtin_readcyclecounter() - t_loop; d[4] = ncols; d[5] = 1;
	}
}

// ---- X runs (round 5): the column loop of a Y-form run WITHOUT memory operations on the plain path ---------------------------------------
// Round 4's stamps: a plain column of slot_run_body takes 230 - 250 cycles for 12 useful instructions -- four LDS reads per column requested
// ahead, a wait the compiler places at the head of the four-column trip, the thread's operand rebuilt from two LDS words, a control word
// through v_readlane, loop-carried LDS pointers merged at every break.  Measured on the device (scripts/micro/r5_probe.hip): four v_sad_u32,
// a scalar test and a branch issue in 44 cycles; a scalar-cache hit costs 72 cycles, not 300; straight-line code runs as fast as a loop.
// So a run of at most SLOT_XCOLS columns keeps, per thread, the operand X0 of EVERY column in a register of its own (built once in the
// prologue from the create-time tables: lane part + workgroup part + wave part), the run's control words in 16 SGPRs, and the wave-uniform
// Kr words of four columns at a time in 16 SGPRs fetched through the scalar cache one trip ahead (tab_kr: contiguous per run) -- a plain
// column is  4 x v_sad_u32 (VGPR, SGPR, VGPR) + s_and + s_cbranch  and nothing else: no LDS, no wait, no address arithmetic.  The columns are
// evaluated in pairs of trips of four; the columns behind the run's last are zero everywhere (harmless), so the loop is counted and has no way out in the middle.
// An ending read: the tie parity of the thread under the read's mask is ONE bit of a per-thread word built at create time (tab_par: bit e for
// the run's e-th ending read), so the kernel needs neither the mask nor a popcount; the partner cells are requested FIRST (ds_bpermute / LDS
// exchange), the lane masks of the four cells are prepared on the scalar unit while they travel, one wait, then the decisions
// (borrow / carry chain, kernels above) and the maxima: hand-written blocks (slotx_*), every wait count and hazard distance written out.
// LDS holds the wave-slot exchange buffers only; the prologue has no barrier.
typedef uint32_t slot_u32x4 __attribute__((ext_vector_type(4)));

// ONE ending read of an X run, Y form, four cells per thread -- a single hand-scheduled block (the compiler sees no control flow and no LDS
// operation of it; every wait and every hazard distance is written out):
//   1. the partner cells are REQUESTED first, into v60 .. v63: ds_bpermute (lane slot: lane ^ 2^(slot - 2)), the 16 bytes of the partner wave's thread
//      through LDS + barrier (wave slot; D travels as v[56:59]), or the thread's own cells (reg slot);
//   2. while they travel, the scalar unit prepares the lane masks Q_r of the four cells: the thread's tie parity is bit e of `par` (tab_par), cell r flips it
//      by bit r of qmask -- bit 0 of a qmask is never set (the planner asserts it), so Q_0 is the thread mask itself;
//   3. one wait; the four borrows  Y_r - O_r - q_r  (v_subb with Q_r as carry-in) are formed first and consumed in the same order -- on gfx950 a VALU
//      result in an SGPR may be read by a VALU two instructions later at the earliest -- and shifted into the record byte; the maxima in between.
// The block reads the ending read's 16-bit control field (n_end | slot << 2 | qmask << 7 | exchange buffer << 15) from a scalar register; the thread's tie
// parities move down one bit per ending read; the record byte is stored from inside (offset register advanced by the workgroup's size).  Two exchange
// buffers: the barrier of the next exchange also protects this one's reads.
#define SLOTX_ENDING_ASM                                                                                               \
	"s_bfe_u32 %[sl], %[cw], 0x50002\n\t"          /* slot of the ending read: bits 2 .. 6 of its control field */     \
	"s_cmp_lt_u32 %[sl], 8\n\t"                                                                                        \
	"s_cbranch_scc0 .Lxw%=\n\t"                                                                                        \
	"s_cmp_lt_u32 %[sl], 2\n\t"                                                                                        \
	"s_cbranch_scc1 .Lxr%=\n\t"                                                                                        \
	"s_lshl_b32 %[sa], 1, %[sl]\n\t"               /* lane slot: byte address of lane ^ 2^(slot - 2) = (lane * 4) ^ 2^slot */ \
	"v_xor_b32_e32 %[a], %[sa], %[l4]\n\t"                                                                             \
	"ds_bpermute_b32 v60, %[a], %[d0]\n\t"                                                                             \
	"ds_bpermute_b32 v61, %[a], %[d1]\n\t"                                                                             \
	"ds_bpermute_b32 v62, %[a], %[d2]\n\t"                                                                             \
	"ds_bpermute_b32 v63, %[a], %[d3]\n\t"                                                                             \
	"s_branch .Lxm%=\n"                                                                                                \
	".Lxw%=:\n\t"                                  /* wave slot: partner thread = tid ^ (64 << (slot - 8)), 16 bytes each */ \
	"s_sub_u32 %[sa], %[sl], 8\n\t"                                                                                    \
	"s_lshl_b32 %[sa], 0x400, %[sa]\n\t"                                                                               \
	"s_bfe_u32 %[sl], %[cw], 0x1000f\n\t"          /* which of the two exchange buffers: bit 15 of the control field (the planner counts the exchanges) */ \
	"s_mul_i32 %[sl], %[sl], %[xby]\n\t"                                                                               \
	"v_mov_b32_e32 v56, %[d0]\n\t"                                                                                     \
	"v_mov_b32_e32 v57, %[d1]\n\t"                                                                                     \
	"v_mov_b32_e32 v58, %[d2]\n\t"                                                                                     \
	"v_mov_b32_e32 v59, %[d3]\n\t"                                                                                     \
	"v_add_u32_e32 %[a], %[sl], %[t16]\n\t"                                                                            \
	"ds_write_b128 %[a], v[56:59]\n\t"                                                                                 \
	"v_xor_b32_e32 %[a], %[sa], %[t16]\n\t"                                                                            \
	"v_add_u32_e32 %[a], %[sl], %[a]\n\t"                                                                              \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"s_barrier\n\t"                                                                                                    \
	"ds_read_b128 v[60:63], %[a]\n\t"                                                                                  \
	"s_branch .Lxm%=\n"                                                                                                \
	".Lxr%=:\n\t"                                  /* reg slot: the partners are the thread's own cells */             \
	"s_cmp_eq_u32 %[sl], 0\n\t"                                                                                        \
	"s_cbranch_scc0 .Lxq%=\n\t"                                                                                        \
	"v_mov_b32_e32 v60, %[d1]\n\t"                                                                                     \
	"v_mov_b32_e32 v61, %[d0]\n\t"                                                                                     \
	"v_mov_b32_e32 v62, %[d3]\n\t"                                                                                     \
	"v_mov_b32_e32 v63, %[d2]\n\t"                                                                                     \
	"s_branch .Lxm%=\n"                                                                                                \
	".Lxq%=:\n\t"                                                                                                      \
	"v_mov_b32_e32 v60, %[d2]\n\t"                                                                                     \
	"v_mov_b32_e32 v61, %[d3]\n\t"                                                                                     \
	"v_mov_b32_e32 v62, %[d0]\n\t"                                                                                     \
	"v_mov_b32_e32 v63, %[d1]\n"                                                                                       \
	".Lxm%=:\n\t"                                  /* the lane masks, while the partner cells travel: the thread's tie parity is bit 0 of par */ \
	"v_and_b32_e32 %[t], 1, %[par]\n\t"                                                                                \
	"v_cmp_ne_u32_e64 %[qt], 0, %[t]\n\t"                                                                              \
	"v_lshrrev_b32_e32 %[par], 1, %[par]\n\t"      /* (the next ending read's bit moves down) */                       \
	"s_not_b64 %[qn], %[qt]\n\t"                                                                                       \
	"s_bitcmp1_b32 %[cw], 8\n\t"                   /* qmask: bits 7 .. 10 of the control field; bit r flips cell r */  \
	"s_cselect_b64 %[q1], %[qn], %[qt]\n\t"                                                                            \
	"s_bitcmp1_b32 %[cw], 9\n\t"                                                                                       \
	"s_cselect_b64 %[q2], %[qn], %[qt]\n\t"                                                                            \
	"s_bitcmp1_b32 %[cw], 10\n\t"                                                                                      \
	"s_cselect_b64 %[q3], %[qn], %[qt]\n\t"                                                                            \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"v_subb_co_u32_e64 %[t], %[q3], %[d3], v63, %[q3]\n\t"        /* (a borrow overwrites its own carry-in mask) */                                                             \
	"v_subb_co_u32_e64 %[t], %[q2], %[d2], v62, %[q2]\n\t"                                                             \
	"v_subb_co_u32_e64 %[t], %[q1], %[d1], v61, %[q1]\n\t"                                                             \
	"v_subb_co_u32_e64 %[t], %[qt], %[d0], v60, %[qt]\n\t"                                                             \
	"v_cndmask_b32_e64 %[tk], 0, 1, %[q3]\n\t"                                                                         \
	"v_max_u32_e32 %[d3], %[d3], v63\n\t"                                                                              \
	"v_addc_co_u32_e64 %[tk], %[qn], %[tk], %[tk], %[q2]\n\t"                                                          \
	"v_max_u32_e32 %[d2], %[d2], v62\n\t"                                                                              \
	"v_addc_co_u32_e64 %[tk], %[qn], %[tk], %[tk], %[q1]\n\t"                                                          \
	"v_max_u32_e32 %[d1], %[d1], v61\n\t"                                                                              \
	"v_addc_co_u32_e64 %[tk], %[qn], %[tk], %[tk], %[qt]\n\t"                                                          \
	"v_max_u32_e32 %[d0], %[d0], v60\n\t"                                                                              \
	"global_store_byte %[ro], %[tk], %[rb]\n\t"    /* the record byte of this thread and ending read: fire and forget */ \
	"v_add_u32_e32 %[ro], %[thr], %[ro]"

// (prologue + column loop of an X run: leaves the thread's four cells in D; the exit -- slot_runx_exit -- is a call of its own so that a kernel whose run
//  descriptor lives in memory can fetch the exit's half of it AFTER the loop instead of carrying ~30 scalars through it)
// One word of each of the N 64-byte lines behind `base` requested into ONE scalar register (the values are never read; loads that return in any order into the
// same register are harmless): the lines are in the scalar cache when the loop asks for them.  The register stays reserved -- slot_touch_done -- until a wait
// for scalar data has passed, or the compiler would hand it to a live value that a late return then overwrites.
#define SLOT_TOUCH_1(o) "s_load_dword %0, %1, " #o "\n\t"
#define SLOT_TOUCH_4(a, b, c, d) SLOT_TOUCH_1(a) SLOT_TOUCH_1(b) SLOT_TOUCH_1(c) SLOT_TOUCH_1(d)
template <int N>
__device__ __forceinline__ uint32_t slot_touch_lines(const void* base) {   // lines 1 .. N (line 0 is what the prologue loads anyway)
	static_assert(N == 2 || N == 4 || N == 8 || N == 12 || N == 16, "touch counts that are written out");
	uint32_t junk;
	const unsigned long long b = (unsigned long long)base;
	if (N == 2) asm volatile(SLOT_TOUCH_1(64) SLOT_TOUCH_1(128) : "=s"(junk) : "s"(b) : "memory");
	if (N == 4) asm volatile(SLOT_TOUCH_4(64, 128, 192, 256) : "=s"(junk) : "s"(b) : "memory");
	if (N == 8) asm volatile(SLOT_TOUCH_4(64, 128, 192, 256) SLOT_TOUCH_4(320, 384, 448, 512) : "=s"(junk) : "s"(b) : "memory");
	if (N == 12) asm volatile(SLOT_TOUCH_4(64, 128, 192, 256) SLOT_TOUCH_4(320, 384, 448, 512) SLOT_TOUCH_4(576, 640, 704, 768) : "=s"(junk) : "s"(b) : "memory");
	if (N == 16) asm volatile(SLOT_TOUCH_4(64, 128, 192, 256) SLOT_TOUCH_4(320, 384, 448, 512) SLOT_TOUCH_4(576, 640, 704, 768) SLOT_TOUCH_4(832, 896, 960, 1024) : "=s"(junk) : "s"(b) : "memory");
	return junk;
}
__device__ __forceinline__ void slot_touch_done(uint32_t junk) { asm volatile("" ::"s"(junk)); }

struct SlotxStamps { unsigned long long t_start, t_issued, t_loaded, t_loop; };
template <int LR, int XC, bool DBG>
__device__ __forceinline__ void slot_runx_core(const DevProblem& P, const SlotRun& run, const uint32_t* __restrict__ prev, const uint32_t w, uint32_t (&D)[1 << LR],
                                               SlotxStamps& stamps, const void* warm = nullptr) {
	static_assert(LR == 2, "X runs: four cells per thread (the masks, the decisions and the exchange are written for four)");
	static_assert(XC % 4 == 0 && XC <= SLOT_XCOLS, "whole trips of four columns");
	constexpr bool STREAM = XC == 0;   // the operands of a trip are formed a trip ahead from the tables (no LDS lines: several workgroups per CU -- shared launches)
	constexpr int R = 1 << LR;
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // wave-slot exchange 2 x [threads][R] | X0 [trip][thread][4 columns] (slotx_lds_bytes)
	const unsigned long long t_start = (DBG && P.dbg) ? __builtin_readcyclecounter() : 0ull;
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t wave = uni(tid >> 6);
	const uint32_t L = run.L;
	const uint32_t lthr = tid << LR;
	const uint32_t Pthr = (w << L) | lthr;
	const uint32_t ncols = run.ncols, threads = run.threads;
	const uint32_t ncp = (ncols + 7u) & ~7u;   // row length of the tables G, W and SL of an X run: padded with zeros to a pair of trips (slot_tables)
	const uint32_t* __restrict__ tab = P.slot_tab;
	// ---- prologue: one batch of loads.  Wave-uniform data through the scalar cache: the control words and the Kr words of the first trip, the
	// workgroup's half of the tie parities.
	const uint32_t* __restrict__ kr_tab = tab + run.tab_kr;
	const uint32_t* __restrict__ cw_tab = kr_tab + ((ncols + (uint32_t)SLOT_XPAD) << LR);   // one control word per column, behind the Kr words (slot_tables)
	typedef uint32_t slot_u32x4c __attribute__((ext_vector_type(4)));
	typedef const __attribute__((address_space(4))) slot_u32x4c* slot_cptr4c;
	slot_u32x16 krA = *(slot_cptr16)(unsigned long long)kr_tab, krB;
	slot_u32x4c cwA = *(slot_cptr4c)(unsigned long long)cw_tab, cwB;
	const uint32_t par_w = *(const __attribute__((address_space(4))) uint32_t*)(unsigned long long)(tab + run.tab_par + threads + w);
	// The Kr words of a trip are one 64-byte line, requested a trip ahead -- a MISS of the scalar cache every trip, and scalar loads share their counter with LDS
	// operations: an ending read in the first columns of a trip waited for it with its own `s_waitcnt lgkmcnt(0)` (stamps: 500 - 600 cycles there against 400 in a
	// trip's last column).  The lines of the whole run are touched here, under the prologue's long waits, so that the loop's requests HIT (72 cycles).  One word of
	// each line into a register that stays reserved until the first wait for scalar data below (the loads return in any order; the compiler must not reuse the
	// registers before they have).  Lines behind a short run's last belong to the tables that follow (slot_tables leaves room behind the last run).
	uint32_t touch_kr = 0, touch_kr2 = 0, touch_cw = 0, touch_g = 0, touch_w = 0;
	// the entering cells first (they come from the other XCDs' stores: the longest latency of the prologue) ...
	// Shared launches read their run descriptor from memory (SlotBatchEntry) BEFORE anything else can be requested: a miss all the way to HBM per launch.
	// The entry of the table's next step lies behind this one; its lines are requested here, just before the entering cells (which miss to HBM anyway) --
	// into this XCD's L2, where the next launch's workgroup (x, y) finds them.  The destination register stays reserved until the entering cells are used
	// below: vector loads return in order, so by then this request has landed (the compiler cannot see a load inside an asm statement).
	uint32_t warm_junk = 0;
	if (warm) {
		const unsigned long long line = (unsigned long long)warm + ((lane & 7u) << 6);
		asm volatile("global_load_dword %0, %1, off" : "=v"(warm_junk) : "v"(line) : "memory");
	}
	uint32_t Draw[R];
	bool flip;
	slot_enter_cells<LR, DBG>(P, run, prev, Pthr, Draw, flip);
	const uint32_t par_l = tab[run.tab_par + tid];
	const uint32_t* __restrict__ sl_src = tab + run.tab_sl + lane;   // the lane part of column c: sl_src[64 c] (one coalesced 256-byte request per wave and column)
	const uint32_t* __restrict__ g_row = tab + run.tab_g + w * ncp;  // the workgroup's and the wave's part: rows of the tables G and W
	const uint32_t* __restrict__ w_row = tab + run.tab_w + wave * ncp;
	typedef uint32_t slot_u32x4s __attribute__((ext_vector_type(4)));
	typedef const __attribute__((address_space(4))) slot_u32x4s* slot_cptr4;
	slot_u32x4s gA, gB, wA, wB;
	uint32_t slA[4], slB[4];
	unsigned long long t_issued = 0ull;
	if (STREAM) {
		gA = *(slot_cptr4)(unsigned long long)g_row;
		wA = *(slot_cptr4)(unsigned long long)w_row;
#pragma unroll
		for (int k = 0; k < 4; ++k) slA[k] = sl_src[64 * k];
		// (shared launches: the scalar lines of the whole run -- Kr and control words, the workgroup's and the wave's row of G / W -- touched behind the prologue's requests)
		touch_kr = slot_touch_lines<8>(kr_tab);
		touch_cw = slot_touch_lines<2>(cw_tab);
		touch_g = slot_touch_lines<2>(g_row);
		touch_w = slot_touch_lines<2>(w_row);
		t_issued = (DBG && P.dbg) ? __builtin_readcyclecounter() : 0ull;
	} else {
		// lane c of every wave fetches the workgroup + wave part of column c, the thread the lane part of every column (columns behind the run's last
		// read zeros, further ones the tables that follow -- never used)
		const uint32_t cl = lane < ncols ? lane : 0u;
		const uint32_t a_g = g_row[cl], a_w = w_row[cl];
		uint32_t X[XC > 0 ? XC : 4];
#pragma unroll
		for (int c = 0; c < XC; ++c) X[c] = sl_src[64 * c];
		// (issued HERE, behind every request of the prologue: in front of the entering cells' loads the scalar waits of their index arithmetic would have
		// waited for these misses first)
		touch_kr = slot_touch_lines<(XC <= 8 ? 4 : 8)>(kr_tab);   // (XC / 4 + 2 lines of Kr words; a short run's touches reach into the tables behind it: harmless)
		if (XC > 24) touch_kr2 = slot_touch_lines<2>(kr_tab + 128);   // (lines 9, 10 of a 32-column run)
		touch_cw = slot_touch_lines<2>(cw_tab);   // (... and the control words': sixteen columns per line)
		t_issued = (DBG && P.dbg) ? __builtin_readcyclecounter() : 0ull;
		// X0 of column c = 2 A(thread's cell 0) + bias = lane part + (workgroup + wave part, held by lane c): kept in the thread's OWN 16 bytes per trip of
		// an LDS area (nobody else reads them: no barrier) -- a register per column would need the column loop unrolled over the whole run
		const uint32_t Avec = lane < ncols ? a_g + a_w : 0u;   // (columns behind the run's last: zero)
		uint4* __restrict__ xs = reinterpret_cast<uint4*>(smem + 2u * threads * R) + tid;   // + trip * threads  (behind the two exchange buffers)
#pragma unroll
		for (int t4 = 0; t4 < XC / 4; ++t4) {
			uint32_t x[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) x[k] = X[4 * t4 + k] + (uint32_t)__builtin_amdgcn_readlane((int)Avec, 4 * t4 + k);
			xs[(uint32_t)t4 * threads] = make_uint4(x[0], x[1], x[2], x[3]);
		}
	}
	uint32_t par = par_l ^ par_w;   // bit 0: the next ending read
	uint8_t* __restrict__ rec = P.bt + (((unsigned long long)run.rec_hi << 32) | run.rec_lo) + (size_t)w * run.n_ends * threads;   // (wave-uniform; the thread adds tid)
	// D: Y = B - 2 D
#pragma unroll
	for (int r = 0; r < R; ++r) D[r] = flip ? Draw[R - 1 - r] : Draw[r];
	if (!(run.yflags & 2u)) {   // the entering column is in D form
#pragma unroll
		for (int r = 0; r < R; ++r) D[r] = run.base_in - 2u * D[r];
	}
	if (warm) asm volatile("" ::"v"(warm_junk), "v"(D[0]));   // (the entering cells are here, so the request issued before them is too)
	if (DBG && P.dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	const unsigned long long t_loaded = (DBG && P.dbg) ? __builtin_readcyclecounter() + (D[0] & 0u) : 0ull;

	const uint32_t lds0 = (uint32_t)(unsigned long long)smem;   // LDS byte offset of the exchange buffers
	const uint32_t lane4 = lane << 2, tid16 = lds0 + (tid << 4);   // (tid16: this thread's 16 bytes of exchange buffer 0)
	const uint32_t xbytes = threads * R * 4u;                   // one exchange buffer
	uint32_t rec_off = tid;                                     // this thread's byte of the current ending read, relative to rec
	const SlotRow* __restrict__ rows = P.slot_rows + run.row_off;
	// one ending read (its control field is a scalar): SLOTX_ENDING_ASM
	auto ending = [&](const uint32_t field) {
		unsigned long long q1, q2, q3, qt, qn;
		uint32_t t, a, sa, sl, takes;
		asm volatile(SLOTX_ENDING_ASM
		             : [d0] "+v"(D[0]), [d1] "+v"(D[1]), [d2] "+v"(D[2]), [d3] "+v"(D[3]), [par] "+v"(par), [ro] "+v"(rec_off), [tk] "=&v"(takes), [t] "=&v"(t), [a] "=&v"(a),
		               [sa] "=&s"(sa), [sl] "=&s"(sl), [q1] "=&s"(q1), [q2] "=&s"(q2), [q3] "=&s"(q3), [qt] "=&s"(qt), [qn] "=&s"(qn)
		             : [cw] "s"(field), [xby] "s"(xbytes), [thr] "s"(threads), [rb] "s"(rec), [l4] "v"(lane4), [t16] "v"(tid16)
		             : "memory", "scc", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
	};
	// column k of the current trip: its operand x, its four Kr words, its 16 control bits
	uint32_t ci = 0;
	auto column = [&](const uint32_t x, const uint32_t k0, const uint32_t k1, const uint32_t k2, const uint32_t k3, const uint32_t ctrl) {
		D[0] = slot_y_sad_s(x, k0, D[0]);
		D[1] = slot_y_sad_s(x, k1, D[1]);
		D[2] = slot_y_sad_s(x, k2, D[2]);
		D[3] = slot_y_sad_s(x, k3, D[3]);
		if (DBG && P.dbg && w == 0 && tid == 0 && ci < 32u) P.dbg[(size_t)run.pad * 48 + 8 + ci] = __builtin_readcyclecounter() - t_loaded;
		if (ctrl & 3u) {   // reads ending here: the first two are described in the control word, a third and later ones in the row (scalar loads, requested
			              // before the second ending read is evaluated)
			ending(ctrl);
			if (ctrl & 2u) {
				const unsigned long long row = (unsigned long long)(rows + ci);
				uint32_t n_end = 2u, info = 0u;
				if (ctrl & 1u) {   // (a count of 3 says "three or more")
					n_end = *(const __attribute__((address_space(4))) uint32_t*)(row + 44);
					info = *(const __attribute__((address_space(4))) uint32_t*)(row + 64);
				}
				ending(ctrl >> 16);
				for (uint32_t e = 2; e < n_end; ++e) {
					const uint32_t field = ((info & 31u) << 2) | (((info >> 8) & 15u) << 7) | (((info >> 16) & 1u) << 15);   // info: slot | qmask << 8 | exchange buffer << 16
					if (e + 1u < n_end) info = *(const __attribute__((address_space(4))) uint32_t*)(row + 56 + 8 * e);     // (the next one's, a read ahead)
					ending(field);
				}
			}
		}
		++ci;
	};
	// Two trips per loop iteration (no register copies): while trip t is evaluated from set A, the operands, Kr words and control words of trip
	// t + 1 travel into set B (one ds_read_b128 of the thread's own line, two scalar-cache loads), and the other way round.  Scalar loads return
	// out of order, so a wait for set A would also wait for the requests of set B issued before it: the empty statements below USE set A first --
	// the compiler's wait lands in front of them, before the next requests go out.
	// A counted loop without a way out in the middle (a `break` per column made the compiler merge the loop's scalars with undefined values on the
	// exit edge: five VALU -> SGPR copies per column): the columns behind the run's last -- up to seven, to a whole pair of trips -- are HARMLESS,
	// their operands and Kr words are zero (slot_tables pads the lane parts and the Kr table, lanes >= ncols of Avec are zero: |0 - 0| = 0) and no
	// read ends in them (the control words behind the run are zero).
	if (STREAM) {
		{   // (the first wait for scalar data: the touches of the prologue have returned, their registers are free)
			asm volatile("" ::"s"(krA[0]), "s"(cwA[0]), "s"(gA[0]), "s"(wA[0]));
			slot_touch_done(touch_kr); slot_touch_done(touch_cw); slot_touch_done(touch_g); slot_touch_done(touch_w);
		}
		// the operands of a trip: lane part (four coalesced loads, a trip ahead) + the workgroup's and the wave's part (two scalar-cache loads of four words)
		for (uint32_t pairs = (ncols + 7u) >> 3; pairs; --pairs) {
			asm volatile("" ::"s"(krA[0]), "s"(cwA[0]), "s"(gA[0]), "s"(wA[0]));
			krB = *(slot_cptr16)(unsigned long long)(kr_tab + 16u);
			cwB = *(slot_cptr4c)(unsigned long long)(cw_tab + 4u);
			gB = *(slot_cptr4)(unsigned long long)(g_row + 4u);
			wB = *(slot_cptr4)(unsigned long long)(w_row + 4u);
#pragma unroll
			for (int k = 0; k < 4; ++k) slB[k] = sl_src[64 * (4 + k)];
			column(slA[0] + (gA[0] + wA[0]), krA[0], krA[1], krA[2], krA[3], cwA[0]);
			column(slA[1] + (gA[1] + wA[1]), krA[4], krA[5], krA[6], krA[7], cwA[1]);
			column(slA[2] + (gA[2] + wA[2]), krA[8], krA[9], krA[10], krA[11], cwA[2]);
			column(slA[3] + (gA[3] + wA[3]), krA[12], krA[13], krA[14], krA[15], cwA[3]);
			asm volatile("" ::"s"(krB[0]), "s"(cwB[0]), "s"(gB[0]), "s"(wB[0]));
			kr_tab += 32u;
			cw_tab += 8u;
			g_row += 8u;
			w_row += 8u;
			sl_src += 512u;
			krA = *(slot_cptr16)(unsigned long long)kr_tab;
			cwA = *(slot_cptr4c)(unsigned long long)cw_tab;
			gA = *(slot_cptr4)(unsigned long long)g_row;
			wA = *(slot_cptr4)(unsigned long long)w_row;
#pragma unroll
			for (int k = 0; k < 4; ++k) slA[k] = sl_src[64 * k];
			column(slB[0] + (gB[0] + wB[0]), krB[0], krB[1], krB[2], krB[3], cwB[0]);
			column(slB[1] + (gB[1] + wB[1]), krB[4], krB[5], krB[6], krB[7], cwB[1]);
			column(slB[2] + (gB[2] + wB[2]), krB[8], krB[9], krB[10], krB[11], cwB[2]);
			column(slB[3] + (gB[3] + wB[3]), krB[12], krB[13], krB[14], krB[15], cwB[3]);
		}
	} else {
	const uint32_t xstride = threads * 16u;
	uint32_t xaddr = 2u * xstride + tid16;   // LDS byte address of the thread's line of trip 0 (behind the two exchange buffers)
	typedef __attribute__((address_space(3))) const slot_u32x4* lds_line;
	slot_u32x4 xA = *(lds_line)(size_t)xaddr, xB;
	{   // (the first use of the trip's scalars: the compiler's wait for them lands here and covers the touches of the prologue -- their registers are free from here on)
		asm volatile("" ::"s"(krA[0]), "s"(cwA[0]));
		slot_touch_done(touch_kr); slot_touch_done(touch_kr2); slot_touch_done(touch_cw);
	}
	for (uint32_t pairs = (ncols + 7u) >> 3; pairs; --pairs) {
		asm volatile("" ::"s"(krA[0]), "s"(cwA[0]), "v"(xA.x));
		krB = *(slot_cptr16)(unsigned long long)(kr_tab + 16u);
		cwB = *(slot_cptr4c)(unsigned long long)(cw_tab + 4u);
		xB = *(lds_line)(size_t)(xaddr + xstride);
		column(xA.x, krA[0], krA[1], krA[2], krA[3], cwA[0]);
		column(xA.y, krA[4], krA[5], krA[6], krA[7], cwA[1]);
		column(xA.z, krA[8], krA[9], krA[10], krA[11], cwA[2]);
		column(xA.w, krA[12], krA[13], krA[14], krA[15], cwA[3]);
		asm volatile("" ::"s"(krB[0]), "s"(cwB[0]), "v"(xB.x));
		kr_tab += 32u;
		cw_tab += 8u;
		xaddr += 2u * xstride;
		krA = *(slot_cptr16)(unsigned long long)kr_tab;
		cwA = *(slot_cptr4c)(unsigned long long)cw_tab;
		xA = *(lds_line)(size_t)xaddr;
		column(xB.x, krB[0], krB[1], krB[2], krB[3], cwB[0]);
		column(xB.y, krB[4], krB[5], krB[6], krB[7], cwB[1]);
		column(xB.z, krB[8], krB[9], krB[10], krB[11], cwB[2]);
		column(xB.w, krB[12], krB[13], krB[14], krB[15], cwB[3]);
	}
	}
	stamps.t_start = t_start; stamps.t_issued = t_issued; stamps.t_loaded = t_loaded;
	stamps.t_loop = (DBG && P.dbg) ? __builtin_readcyclecounter() + (D[0] & 0u) : 0ull;
}
// ---- the same for EIGHT cells per thread (reg slots 0 .. 2: the layout of wide tables that share their launches, slot_plan.cpp `shared_launches`) ----
// The partner cells travel in v56 .. v63, a wave-slot exchange is 32 bytes per thread (D as v[48:55]), seven lane masks (bit r of qmask flips cell r; bit 0 is
// never set), eight borrows formed first and consumed in the same order.  Streamed operands only (several workgroups per CU): a trip is TWO columns -- their
// sixteen Kr words are one scalar-cache load.
#define SLOTX8_MOVS(a0, a1, a2, a3, a4, a5, a6, a7)                                                                    \
	"v_mov_b32_e32 v56, %[d" #a0 "]\n\tv_mov_b32_e32 v57, %[d" #a1 "]\n\tv_mov_b32_e32 v58, %[d" #a2 "]\n\tv_mov_b32_e32 v59, %[d" #a3 "]\n\t" \
	"v_mov_b32_e32 v60, %[d" #a4 "]\n\tv_mov_b32_e32 v61, %[d" #a5 "]\n\tv_mov_b32_e32 v62, %[d" #a6 "]\n\tv_mov_b32_e32 v63, %[d" #a7 "]\n"
#define SLOTX8_ENDING_ASM                                                                                              \
	"s_bfe_u32 %[sl], %[cw], 0x50002\n\t"          /* slot of the ending read */                                       \
	"s_cmp_lt_u32 %[sl], 9\n\t"                                                                                        \
	"s_cbranch_scc0 .Lyw%=\n\t"                                                                                        \
	"s_cmp_lt_u32 %[sl], 3\n\t"                                                                                        \
	"s_cbranch_scc1 .Lyr%=\n\t"                                                                                        \
	"s_sub_u32 %[sa], %[sl], 1\n\t"                /* lane slot: byte address of lane ^ 2^(slot - 3) = (lane * 4) ^ 2^(slot - 1) */ \
	"s_lshl_b32 %[sa], 1, %[sa]\n\t"                                                                                   \
	"v_xor_b32_e32 %[t], %[sa], %[l4]\n\t"                                                                             \
	"ds_bpermute_b32 v56, %[t], %[d0]\n\t"                                                                             \
	"ds_bpermute_b32 v57, %[t], %[d1]\n\t"                                                                             \
	"ds_bpermute_b32 v58, %[t], %[d2]\n\t"                                                                             \
	"ds_bpermute_b32 v59, %[t], %[d3]\n\t"                                                                             \
	"ds_bpermute_b32 v60, %[t], %[d4]\n\t"                                                                             \
	"ds_bpermute_b32 v61, %[t], %[d5]\n\t"                                                                             \
	"ds_bpermute_b32 v62, %[t], %[d6]\n\t"                                                                             \
	"ds_bpermute_b32 v63, %[t], %[d7]\n\t"                                                                             \
	"s_branch .Lym%=\n"                                                                                                \
	".Lyw%=:\n\t"                                  /* wave slot: partner thread = tid ^ (64 << (slot - 9)), 32 bytes each */ \
	"s_sub_u32 %[sa], %[sl], 9\n\t"                                                                                    \
	"s_lshl_b32 %[sa], 0x800, %[sa]\n\t"                                                                               \
	"s_bfe_u32 %[sl], %[cw], 0x1000f\n\t"          /* which of the two exchange buffers: bit 15 of the control field */ \
	"s_mul_i32 %[sl], %[sl], %[xby]\n\t"                                                                               \
	"v_mov_b32_e32 v48, %[d0]\n\tv_mov_b32_e32 v49, %[d1]\n\tv_mov_b32_e32 v50, %[d2]\n\tv_mov_b32_e32 v51, %[d3]\n\t" \
	"v_mov_b32_e32 v52, %[d4]\n\tv_mov_b32_e32 v53, %[d5]\n\tv_mov_b32_e32 v54, %[d6]\n\tv_mov_b32_e32 v55, %[d7]\n\t" \
	"v_add_u32_e32 %[t], %[sl], %[t32]\n\t"                                                                            \
	"ds_write_b128 %[t], v[48:51]\n\t"                                                                                 \
	"ds_write_b128 %[t], v[52:55] offset:16\n\t"                                                                       \
	"v_xor_b32_e32 %[t], %[sa], %[t32]\n\t"                                                                            \
	"v_add_u32_e32 %[t], %[sl], %[t]\n\t"                                                                              \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"s_barrier\n\t"                                                                                                    \
	"ds_read_b128 v[56:59], %[t]\n\t"                                                                                  \
	"ds_read_b128 v[60:63], %[t] offset:16\n\t"                                                                        \
	"s_branch .Lym%=\n"                                                                                                \
	".Lyr%=:\n\t"                                  /* reg slot s: the partner of cell r is cell r ^ 2^s */             \
	"s_cmp_eq_u32 %[sl], 0\n\t"                                                                                        \
	"s_cbranch_scc0 .Lyq%=\n\t" SLOTX8_MOVS(1, 0, 3, 2, 5, 4, 7, 6)                                                    \
	"\ts_branch .Lym%=\n"                                                                                              \
	".Lyq%=:\n\t"                                                                                                      \
	"s_cmp_eq_u32 %[sl], 1\n\t"                                                                                        \
	"s_cbranch_scc0 .Lyp%=\n\t" SLOTX8_MOVS(2, 3, 0, 1, 6, 7, 4, 5)                                                    \
	"\ts_branch .Lym%=\n"                                                                                              \
	".Lyp%=:\n\t" SLOTX8_MOVS(4, 5, 6, 7, 0, 1, 2, 3)                                                                  \
	".Lym%=:\n\t"                                  /* the lane masks, while the partner cells travel */                \
	"v_and_b32_e32 %[t], 1, %[par]\n\t"                                                                                \
	"v_cmp_ne_u32_e64 %[qt], 0, %[t]\n\t"                                                                              \
	"v_lshrrev_b32_e32 %[par], 1, %[par]\n\t"                                                                          \
	"s_not_b64 %[qn], %[qt]\n\t"                                                                                       \
	"s_bitcmp1_b32 %[cw], 8\n\t"                                                                                       \
	"s_cselect_b64 %[q1], %[qn], %[qt]\n\t"                                                                            \
	"s_bitcmp1_b32 %[cw], 9\n\t"                                                                                       \
	"s_cselect_b64 %[q2], %[qn], %[qt]\n\t"                                                                            \
	"s_bitcmp1_b32 %[cw], 10\n\t"                                                                                      \
	"s_cselect_b64 %[q3], %[qn], %[qt]\n\t"                                                                            \
	"s_bitcmp1_b32 %[cw], 11\n\t"                                                                                      \
	"s_cselect_b64 %[q4], %[qn], %[qt]\n\t"                                                                            \
	"s_bitcmp1_b32 %[cw], 12\n\t"                                                                                      \
	"s_cselect_b64 %[q5], %[qn], %[qt]\n\t"                                                                            \
	"s_bitcmp1_b32 %[cw], 13\n\t"                                                                                      \
	"s_cselect_b64 %[q6], %[qn], %[qt]\n\t"                                                                            \
	"s_bitcmp1_b32 %[cw], 14\n\t"                                                                                      \
	"s_cselect_b64 %[q7], %[qn], %[qt]\n\t"                                                                            \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"v_subb_co_u32_e64 %[t], %[q7], %[d7], v63, %[q7]\n\t"        /* (a borrow overwrites its own carry-in mask) */    \
	"v_subb_co_u32_e64 %[t], %[q6], %[d6], v62, %[q6]\n\t"                                                             \
	"v_subb_co_u32_e64 %[t], %[q5], %[d5], v61, %[q5]\n\t"                                                             \
	"v_subb_co_u32_e64 %[t], %[q4], %[d4], v60, %[q4]\n\t"                                                             \
	"v_subb_co_u32_e64 %[t], %[q3], %[d3], v59, %[q3]\n\t"                                                             \
	"v_subb_co_u32_e64 %[t], %[q2], %[d2], v58, %[q2]\n\t"                                                             \
	"v_subb_co_u32_e64 %[t], %[q1], %[d1], v57, %[q1]\n\t"                                                             \
	"v_subb_co_u32_e64 %[t], %[qt], %[d0], v56, %[qt]\n\t"                                                             \
	"v_cndmask_b32_e64 %[tk], 0, 1, %[q7]\n\t"                                                                         \
	"v_max_u32_e32 %[d7], %[d7], v63\n\t"                                                                              \
	"v_addc_co_u32_e64 %[tk], %[qn], %[tk], %[tk], %[q6]\n\t"                                                          \
	"v_max_u32_e32 %[d6], %[d6], v62\n\t"                                                                              \
	"v_addc_co_u32_e64 %[tk], %[qn], %[tk], %[tk], %[q5]\n\t"                                                          \
	"v_max_u32_e32 %[d5], %[d5], v61\n\t"                                                                              \
	"v_addc_co_u32_e64 %[tk], %[qn], %[tk], %[tk], %[q4]\n\t"                                                          \
	"v_max_u32_e32 %[d4], %[d4], v60\n\t"                                                                              \
	"v_addc_co_u32_e64 %[tk], %[qn], %[tk], %[tk], %[q3]\n\t"                                                          \
	"v_max_u32_e32 %[d3], %[d3], v59\n\t"                                                                              \
	"v_addc_co_u32_e64 %[tk], %[qn], %[tk], %[tk], %[q2]\n\t"                                                          \
	"v_max_u32_e32 %[d2], %[d2], v58\n\t"                                                                              \
	"v_addc_co_u32_e64 %[tk], %[qn], %[tk], %[tk], %[q1]\n\t"                                                          \
	"v_max_u32_e32 %[d1], %[d1], v57\n\t"                                                                              \
	"v_addc_co_u32_e64 %[tk], %[qn], %[tk], %[tk], %[qt]\n\t"                                                          \
	"v_max_u32_e32 %[d0], %[d0], v56\n\t"                                                                              \
	"global_store_byte %[ro], %[tk], %[rb]\n\t"                                                                        \
	"v_add_u32_e32 %[ro], %[thr], %[ro]"

template <bool DBG>
__device__ __forceinline__ void slot_runx8_core(const DevProblem& P, const SlotRun& run, const uint32_t* __restrict__ prev, const uint32_t w, uint32_t (&D)[8],
                                                SlotxStamps& stamps, const void* warm = nullptr) {
	constexpr int LR = 3, R = 8;
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // wave-slot exchange 2 x [threads][8]
	const unsigned long long t_start = (DBG && P.dbg) ? __builtin_readcyclecounter() : 0ull;
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t wave = uni(tid >> 6);
	const uint32_t L = run.L;
	const uint32_t Pthr = (w << L) | (tid << LR);
	const uint32_t ncols = run.ncols, threads = run.threads;
	const uint32_t ncp = (ncols + 7u) & ~7u;
	const uint32_t* __restrict__ tab = P.slot_tab;
	const uint32_t* __restrict__ kr_tab = tab + run.tab_kr;               // [column][8]: sixteen words = two columns
	const uint32_t* __restrict__ cw_tab = kr_tab + ((ncols + (uint32_t)SLOT_XPAD) << LR);   // one control word per column, behind the Kr words (slot_tables)
	typedef const __attribute__((address_space(4))) uint32_t* slot_cptr1;
	slot_u32x16 krA = *(slot_cptr16)(unsigned long long)kr_tab, krB;
	slot_u32x2 cwA = *(slot_cptr2)(unsigned long long)cw_tab, cwB;
	const uint32_t par_w = *(slot_cptr1)(unsigned long long)(tab + run.tab_par + threads + w);
	uint32_t warm_junk = 0;
	if (warm) {   // (the next step's entry into this XCD's L2: see slot_runx_core)
		const unsigned long long line = (unsigned long long)warm + ((lane & 7u) << 6);
		asm volatile("global_load_dword %0, %1, off" : "=v"(warm_junk) : "v"(line) : "memory");
	}
	uint32_t Draw[R];
	bool flip;
	slot_enter_cells<LR, DBG>(P, run, prev, Pthr, Draw, flip);
	const uint32_t par_l = tab[run.tab_par + tid];
	const uint32_t* __restrict__ sl_src = tab + run.tab_sl + lane;
	const uint32_t* __restrict__ g_row = tab + run.tab_g + w * ncp;
	const uint32_t* __restrict__ w_row = tab + run.tab_w + wave * ncp;
	slot_u32x2 gA = *(slot_cptr2)(unsigned long long)g_row, gB;
	slot_u32x2 wA = *(slot_cptr2)(unsigned long long)w_row, wB;
	uint32_t slA[2] = {sl_src[0], sl_src[64]}, slB[2];
	// (the scalar lines of the whole run touched behind the prologue's requests, as in slot_runx_core: a trip of TWO columns is one line of Kr words)
	const uint32_t touch_kr = slot_touch_lines<16>(kr_tab), touch_cw = slot_touch_lines<2>(cw_tab), touch_g = slot_touch_lines<2>(g_row), touch_w = slot_touch_lines<2>(w_row);
	const unsigned long long t_issued = (DBG && P.dbg) ? __builtin_readcyclecounter() : 0ull;
	uint32_t par = par_l ^ par_w;
	uint8_t* __restrict__ rec = P.bt + (((unsigned long long)run.rec_hi << 32) | run.rec_lo) + (size_t)w * run.n_ends * threads;
#pragma unroll
	for (int r = 0; r < R; ++r) D[r] = flip ? Draw[R - 1 - r] : Draw[r];
	if (!(run.yflags & 2u)) {
#pragma unroll
		for (int r = 0; r < R; ++r) D[r] = run.base_in - 2u * D[r];
	}
	if (warm) asm volatile("" ::"v"(warm_junk), "v"(D[0]));
	if (DBG && P.dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	const unsigned long long t_loaded = (DBG && P.dbg) ? __builtin_readcyclecounter() + (D[0] & 0u) : 0ull;

	const uint32_t lds0 = (uint32_t)(unsigned long long)smem;
	const uint32_t lane4 = lane << 2, tid32 = lds0 + (tid << 5);
	const uint32_t xbytes = threads * R * 4u;
	uint32_t rec_off = tid;
	const SlotRow* __restrict__ rows = P.slot_rows + run.row_off;
	auto ending = [&](const uint32_t field) {
		unsigned long long q1, q2, q3, q4, q5, q6, q7, qt, qn;
		uint32_t t, sa, sl, takes;
		asm volatile(SLOTX8_ENDING_ASM
		             : [d0] "+v"(D[0]), [d1] "+v"(D[1]), [d2] "+v"(D[2]), [d3] "+v"(D[3]), [d4] "+v"(D[4]), [d5] "+v"(D[5]), [d6] "+v"(D[6]), [d7] "+v"(D[7]), [par] "+v"(par),
		               [ro] "+v"(rec_off), [tk] "=&v"(takes), [t] "=&v"(t), [sa] "=&s"(sa), [sl] "=&s"(sl), [q1] "=&s"(q1), [q2] "=&s"(q2), [q3] "=&s"(q3), [q4] "=&s"(q4),
		               [q5] "=&s"(q5), [q6] "=&s"(q6), [q7] "=&s"(q7), [qt] "=&s"(qt), [qn] "=&s"(qn)
		             : [cw] "s"(field), [xby] "s"(xbytes), [thr] "s"(threads), [rb] "s"(rec), [l4] "v"(lane4), [t32] "v"(tid32)
		             : "memory", "scc", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
	};
	uint32_t ci = 0;
	auto column = [&](const uint32_t x, const slot_u32x16& kr, const int half, const uint32_t ctrl) {
#pragma unroll
		for (int r = 0; r < R; ++r) D[r] = slot_y_sad_s(x, kr[half * 8 + r], D[r]);
		if (DBG && P.dbg && w == 0 && tid == 0 && ci < 32u) P.dbg[(size_t)run.pad * 48 + 8 + ci] = __builtin_readcyclecounter() - t_loaded;
		if (ctrl & 3u) {   // (the first two ending reads in the control word, later ones in the row: as in slot_runx_core)
			ending(ctrl);
			if (ctrl & 2u) {
				const unsigned long long row = (unsigned long long)(rows + ci);
				uint32_t n_end = 2u, info = 0u;
				if (ctrl & 1u) {
					n_end = *(slot_cptr1)(row + 44);
					info = *(slot_cptr1)(row + 64);
				}
				ending(ctrl >> 16);
				for (uint32_t e = 2; e < n_end; ++e) {
					const uint32_t field = ((info & 31u) << 2) | (((info >> 8) & 255u) << 7) | (((info >> 16) & 1u) << 15);
					if (e + 1u < n_end) info = *(slot_cptr1)(row + 56 + 8 * e);
					ending(field);
				}
			}
		}
		++ci;
	};
	{   // (the first wait for scalar data: the touches of the prologue have returned, their registers are free)
		asm volatile("" ::"s"(krA[0]), "s"(cwA[0]), "s"(gA[0]), "s"(wA[0]));
		slot_touch_done(touch_kr); slot_touch_done(touch_cw); slot_touch_done(touch_g); slot_touch_done(touch_w);
	}
	// two trips (of two columns) per loop iteration; the columns behind the run's last are harmless (zeros), as in slot_runx_core
	for (uint32_t quads = (ncols + 3u) >> 2; quads; --quads) {
		asm volatile("" ::"s"(krA[0]), "s"(cwA[0]), "s"(gA[0]), "s"(wA[0]));
		krB = *(slot_cptr16)(unsigned long long)(kr_tab + 16u);
		cwB = *(slot_cptr2)(unsigned long long)(cw_tab + 2u);
		gB = *(slot_cptr2)(unsigned long long)(g_row + 2u);
		wB = *(slot_cptr2)(unsigned long long)(w_row + 2u);
		slB[0] = sl_src[128];
		slB[1] = sl_src[192];
		column(slA[0] + (gA[0] + wA[0]), krA, 0, cwA[0]);
		column(slA[1] + (gA[1] + wA[1]), krA, 1, cwA[1]);
		asm volatile("" ::"s"(krB[0]), "s"(cwB[0]), "s"(gB[0]), "s"(wB[0]));
		kr_tab += 32u;
		cw_tab += 4u;
		g_row += 4u;
		w_row += 4u;
		sl_src += 256u;
		krA = *(slot_cptr16)(unsigned long long)kr_tab;
		cwA = *(slot_cptr2)(unsigned long long)cw_tab;
		gA = *(slot_cptr2)(unsigned long long)g_row;
		wA = *(slot_cptr2)(unsigned long long)w_row;
		slA[0] = sl_src[0];
		slA[1] = sl_src[64];
		column(slB[0] + (gB[0] + wB[0]), krB, 0, cwB[0]);
		column(slB[1] + (gB[1] + wB[1]), krB, 1, cwB[1]);
	}
	stamps.t_start = t_start; stamps.t_issued = t_issued; stamps.t_loaded = t_loaded;
	stamps.t_loop = (DBG && P.dbg) ? __builtin_readcyclecounter() + (D[0] & 0u) : 0ull;
}

template <int LR, bool DBG, bool SPEC>
__device__ __forceinline__ void slot_runx_exit(const DevProblem& P, const SlotRun& run, uint32_t* __restrict__ cur, uint32_t* score_out, const uint32_t w, uint32_t (&D)[1 << LR],
                                               const SlotxStamps& stamps) {
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t wave = uni(tid >> 6);
	const uint32_t L = run.L, lthr = tid << LR;
	slot_exit_cells<LR, DBG, SPEC, true>(P, run, cur, D, w, tid, lane, wave, L, lthr, (w << L) | lthr, run.threads);
	if (score_out && w == 0 && tid == 0) *score_out = D[0];
	if (DBG && P.dbg && w == 0 && tid == 0) {
		unsigned long long* d = P.dbg + (size_t)run.pad * 48;
		d[0] = stamps.t_issued - stamps.t_start; d[1] = stamps.t_loaded - stamps.t_start; d[2] = stamps.t_loop - stamps.t_loaded; d[3] = __builtin_readcyclecounter() - stamps.t_loop;
		d[4] = run.ncols; d[5] = 1;
	}
}
template <int LR, int XC, bool DBG, bool SPEC>
__device__ __forceinline__ void slot_runx_body(const DevProblem& P, const SlotRun& run, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur, const uint32_t w,
                                               uint32_t* score_out) {
	uint32_t D[1 << LR];
	SlotxStamps stamps;
	slot_runx_core<LR, XC, DBG>(P, run, prev, w, D, stamps);
	slot_runx_exit<LR, DBG, SPEC>(P, run, cur, score_out, w, D, stamps);
}

// The per-run tables of the prologue (SlotRun::tab_g / tab_w / tab_sl), once per table at create time: blockIdx.y = run.
__global__ __launch_bounds__(256) void slot_tables(DevProblem P, const SlotRun* __restrict__ runs, uint32_t* __restrict__ tab) {
	const SlotRun& run = runs[blockIdx.y];
	const uint32_t ncols = run.ncols, L = run.L, lr = run.lr, nwg = 1u << (run.g - run.half), nwaves = run.threads >> 6;
	const uint32_t ncp = (run.yflags & 8u) ? (ncols + 7u) & ~7u : ncols;   // X runs: every row padded with zeros to a pair of trips (the columns behind the run's last are harmless)
	const uint32_t n_g = nwg * ncp, n_w = nwaves * ncp, n_sl = ncp * 64u;
	// X runs (slot_runx_body): the Kr words of the run's columns side by side, and the tie parities -- bit e of a thread's word = parity of its local
	// index under the mask of the run's e-th ending read (forward order: by column, then by position in the row), the same for a workgroup's grid bits
	const bool xrun = (run.yflags & 8u) != 0u;
	// ... and behind the Kr words one CONTROL WORD per column: the fields of its first two ending reads (n_end | slot << 2 | qmask << 7 | exchange buffer << 15,
	// the second one in the high half) -- an irregular layout ends two reads in every third column, and fetching the second one's description from the row
	// cost a scalar-cache miss each time
	const uint32_t R = 1u << lr, n_krw = xrun ? (ncols + (uint32_t)SLOT_XPAD) * R : 0u, n_kr = xrun ? n_krw + ncols + (uint32_t)SLOT_XPAD : 0u, n_par = xrun ? run.threads + nwg : 0u;
	const SlotRow* __restrict__ rows = P.slot_rows + run.row_off;
	const uint32_t ysh = run.yflags & 1u, ybias = ysh ? SLOT_YBIAS : 0u;   // Y form: X0 = 2 A + bias = (2 G + bias) + 2 W + 2 SL
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_g + n_w + n_sl + n_kr + n_par; i += gridDim.x * blockDim.x) {
		if (i >= n_g + n_w + n_sl + n_kr) {
			const uint32_t q = i - (n_g + n_w + n_sl + n_kr);
			const uint32_t index = q < run.threads ? (q << lr) : ((q - run.threads) << L);   // a thread's local index / a workgroup's grid bits
			const uint32_t keep = q < run.threads ? (1u << L) - 1u : ~((1u << L) - 1u);
			uint32_t bits = 0, e = 0;
			for (uint32_t c = 0; c < ncols; ++c) {
				const uint32_t n_end = rows[c].n_end;
				for (uint32_t k = 0; k < n_end; ++k, ++e) bits |= ((uint32_t)__popc(index & keep & rows[c].end[k].M) & 1u) << (e & 31u);
			}
			tab[run.tab_par + q] = bits;
		} else if (i >= n_g + n_w + n_sl + n_krw) {
			const uint32_t c = i - (n_g + n_w + n_sl + n_krw);
			uint32_t word = 0;
			if (c < ncols) {
				const SlotRow& row = rows[c];
				auto field = [](uint32_t info) { return ((info & 31u) << 2) | (((info >> 8) & 255u) << 7) | (((info >> 16) & 1u) << 15); };
				if (row.n_end) word = (row.n_end < 3u ? row.n_end : 3u) | field(row.end[0].info);
				if (row.n_end > 1u) word |= field(row.end[1].info) << 16;
			}
			tab[run.tab_kr + n_krw + c] = word;
		} else if (i >= n_g + n_w + n_sl) {
			const uint32_t q = i - (n_g + n_w + n_sl), c = q / R, r = q % R;
			tab[run.tab_kr + q] = c < ncols ? reinterpret_cast<const uint32_t*>(rows + c)[r] : 0u;   // (a Y-form row holds Kr[0 .. R) in its first words)
		} else if (i < n_g) {
			const uint32_t w = i / ncp, c = i % ncp;
			const SlotRow& row = rows[c < ncols ? c : 0u];
			uint32_t acc = row.Cp;
			for (uint32_t s = L; s < L + run.g; ++s) acc += (uint32_t)row.dslot[s] & (0u - ((w >> (s - L)) & 1u));
			tab[run.tab_g + i] = c < ncols ? (acc << ysh) + ybias : 0u;
		} else if (i < n_g + n_w) {
			const uint32_t q = i - n_g, wave = q / ncp, c = q % ncp;
			const SlotRow& row = rows[c < ncols ? c : 0u];
			uint32_t acc = 0;
			for (uint32_t s = lr + 6u; s < L; ++s) acc += (uint32_t)row.dslot[s] & (0u - ((wave >> (s - lr - 6u)) & 1u));
			tab[run.tab_w + q] = c < ncols ? acc << ysh : 0u;
		} else {
			const uint32_t q = i - n_g - n_w, c = q >> 6, lane = q & 63u;
			const SlotRow& row = rows[c < ncols ? c : 0u];
			uint32_t acc = 0;
			for (uint32_t j = 0; j < (uint32_t)SLOT_LANE; ++j) acc += (uint32_t)row.dslot[lr + j] & (0u - ((lane >> j) & 1u));   // (= dlane[j]; a Y-form row reuses those words)
			tab[run.tab_sl + q] = c < ncols ? acc << ysh : 0u;
		}
	}
}

// DBG: timing experiments (WHAMD_SLOT_SKIP switches parts off -- results invalid) and in-kernel cycle stamps (WHAMD_SLOT_STAMPS);
// the production instantiation carries none of it (a single wave issues one VALU instruction per ~8 cycles -- measured,
// scripts/micro/issue_rate.hip -- so every instruction on the column chain counts).
// SPEC: the run ends a backtrace chunk and leaves the seed of the speculative walk (instantiated separately: the other runs
// do not even carry the test).
// One X run as a launch of its own: XC = 24 or 32 unrolled columns (a run of 22 columns does not fetch the lane parts of 32).
// `pack` != 0 (narrow tables: at most 32 workgroups): the launch has EIGHT times the run's workgroups and only every eighth works -- workgroups go to the XCDs
// round robin, so the run's workgroups all sit on XCD 0 and the column they hand to the next launch stays in one L2 (2.26 us instead of 2.89 per dependent
// launch, scripts/micro/r5_boundary.hip; the seven idle workgroups per real one leave at once).
template <int LR, int XC, bool DBG, bool SPEC>
__global__ __launch_bounds__(512) void slot_runx(DevProblem P, SlotRun run, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur, uint32_t* __restrict__ score_out,
                                                 uint32_t pack) {
	if (pack && (blockIdx.x & 7u)) return;
	touch_kernel_arguments<sizeof(DevProblem) + sizeof(SlotRun) + 28>();
	slot_runx_body<LR, XC, DBG, SPEC>(P, run, prev, cur, pack ? blockIdx.x >> 3 : blockIdx.x, score_out);
}

template <int LR, bool DBG, bool SPEC, bool YF = false>
__global__ __launch_bounds__(512) void slot_run(DevProblem P, SlotRun run, const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur,
                                                uint32_t* __restrict__ score_out) {
	touch_kernel_arguments<sizeof(DevProblem) + sizeof(SlotRun) + 24>();
	slot_run_body<LR, DBG, SPEC, YF>(P, run, prev, cur, blockIdx.x, score_out);
}

// One launch = the next run of SEVERAL independent jobs (connected components of one table): blockIdx.y selects the entry.
template <int LR>
__global__ __launch_bounds__(512) void slot_batch(DevProblem P, const SlotBatchEntry* __restrict__ entries) {
	const SlotBatchEntry* __restrict__ e = entries + blockIdx.y;
	{
		const uint32_t* lines = reinterpret_cast<const uint32_t*>(e);
		uint32_t acc = 0;
#pragma unroll
		for (uint32_t l = 0; l < (sizeof(SlotBatchEntry) + 63) / 64; ++l) acc |= lines[l * 16];
		asm volatile("" ::"s"(__builtin_amdgcn_readfirstlane(acc)));
	}
	const SlotRun run = e->run;
	if (blockIdx.x >= (1u << (run.g - run.half)) || threadIdx.x >= run.threads) return;
	if ((LR == 2 || LR == 3) && (run.yflags & 1u)) slot_run_body<(LR == 3 ? 3 : 2), false, false, true>(P, run, e->prev, e->cur, blockIdx.x, e->score_out);
	else slot_run_body<LR, false, false>(P, run, e->prev, e->cur, blockIdx.x, e->score_out);
}

// A wave-uniform record in global memory, copied through the CONSTANT address space: the loads become s_load_dwordx16 (a pointer that was
// itself loaded from memory is not a noalias kernel argument -- through the generic address space the compiler fetches the record with
// vector loads and a v_readfirstlane per word, ~35 cycles each).
template <class T>
__device__ __forceinline__ T slot_scalar_copy(const T* __restrict__ p) {
	static_assert(sizeof(T) % 4 == 0, "whole words");
	T out;
	uint32_t* w = reinterpret_cast<uint32_t*>(&out);
	const __attribute__((address_space(4))) uint32_t* src = (const __attribute__((address_space(4))) uint32_t*)(unsigned long long)p;
#pragma unroll
	for (uint32_t i = 0; i < sizeof(T) / 4; ++i) w[i] = src[i];
	return out;
}

// The arrays of the table an entry belongs to, as the DevProblem the run bodies read (unused fields fold away).
__device__ __forceinline__ DevProblem slot_entry_problem(const SlotBatchEntry& e, bool ped) {
	DevProblem P{};
	if (ped) { P.pslot_tab = e.tab; P.pslot_rows = reinterpret_cast<const PedSlotRow*>(e.rows); }
	else { P.slot_tab = e.tab; P.slot_rows = reinterpret_cast<const SlotRow*>(e.rows); }
	P.slot_ctrl = e.ctrl;
	P.bt = e.bt;
	P.spec_keys = e.spec_keys;
	P.spec_stride = e.spec_stride;
	return P;
}

// Shared launches read their run descriptor from memory before anything else can be requested -- a miss all the way to HBM per launch.  The entry of a
// table's next step lies behind the current one (the schedule's entries are consecutive): the kernel requests its lines with ONE vector load at its very
// start -- into this XCD's L2, where the next launch's workgroup (x, y) finds them (measured: 24 coverage-15 tables 11.1 -> 10.5 us per launch).  The
// destination register must stay reserved until the load has landed (the compiler cannot see a load inside an asm statement): slot_warm_done() at the
// kernel's end -- vector loads return in order, and the body has consumed later ones by then.
__device__ __forceinline__ uint32_t slot_warm_next(const SlotBatchEntry* ep) {
	uint32_t junk;
	const unsigned long long line = (unsigned long long)(ep + 1) + ((threadIdx.x & 7u) << 6);
	asm volatile("global_load_dword %0, %1, off" : "=v"(junk) : "v"(line) : "memory");
	return junk;
}
__device__ __forceinline__ void slot_warm_done(uint32_t junk) { asm volatile("" ::"v"(junk)); }

// Which table and which workgroup of its run a workgroup of a group launch is.  Workgroups go to the XCDs round robin in their linear order (x fastest):
// with the TABLE as x -- `args.pad` != 0, the table dimension padded to a multiple of eight -- every workgroup of a table runs on ONE XCD, and the column a run
// hands to the next stays in that XCD's L2 instead of crossing to another one (scripts/micro/r5_boundary.hip: 2.26 us instead of 2.89 per dependent launch).
// DeviceTable::enqueue_group chooses it where the tables spread evenly over the eight XCDs.
struct SlotGroupWho { uint32_t table, w; bool none; };
__device__ __forceinline__ SlotGroupWho slot_group_who(const SlotGroupArgs& args) {
	SlotGroupWho who;
	who.table = args.pad ? blockIdx.x : blockIdx.y;
	who.w = args.pad ? blockIdx.y : blockIdx.x;
	who.none = who.table >= args.n;
	return who;
}

// The X runs of several tables in one launch (the counterpart of slot_group below for runs that take the X kernel; the group's other runs go out as
// a slot_group launch of their own).  The entry is read TWICE: the prologue's and the loop's half before the loop, the exit's half -- exchange layout,
// masks, the speculative seed's slot -- after it (scalar-cache hits), so the loop's scalar budget is the loop's alone.
template <int LR = 2, bool DBG = false>
__global__ __launch_bounds__(512, 4) void slot_groupx(SlotGroupArgs args) {
	const SlotGroupWho who = slot_group_who(args);
	if (who.none) return;
	const SlotBatchEntry* ep = args.entry[who.table];
	uint32_t D[1 << LR];
	SlotxStamps stamps;
	{
		const SlotBatchEntry e = slot_scalar_copy(ep);
		if (who.w >= (1u << (e.run.g - e.run.half)) || threadIdx.x >= e.run.threads) return;
		DevProblem P = slot_entry_problem(e, false);
		if (DBG) P.dbg_flags = e.pad2;
		const void* warm = (e.pad2 & 0x10000u) ? nullptr : (const void*)(ep + 1);   // (pad2 bit 16: timing experiment, debug library)
		if constexpr (LR == 3) slot_runx8_core<DBG>(P, e.run, e.prev, who.w, D, stamps, warm);
		else slot_runx_core<2, 0, DBG>(P, e.run, e.prev, who.w, D, stamps, warm);
	}
	asm volatile("" : "+s"(ep));   // (opaque: what follows is fetched again, not kept in registers through the loop)
	const SlotBatchEntry e = slot_scalar_copy(ep);
	DevProblem P = slot_entry_problem(e, false);
	if (e.run.spec_id) slot_runx_exit<LR, DBG, true>(P, e.run, e.cur, e.score_out, who.w, D, stamps);
	else slot_runx_exit<LR, DBG, false>(P, e.run, e.cur, e.score_out, who.w, D, stamps);
}

// One launch = the next run of SEVERAL TABLES (whamd_dptable_enqueue_many: independent tables advance in lockstep on one stream):
// blockIdx.y selects the table's entry, blockIdx.x the workgroup of that run.  The launch boundary between two dependent launches
// (~2.5 us) and the prologue's round trips are then paid once per super-step of the whole group instead of once per table, and a
// narrow table (coverage 15: 8 workgroups) no longer leaves the other 248 CUs idle.  Runs that end a backtrace chunk leave their
// seed (SPEC) -- a wave-uniform branch between the two instantiations, not a test on the column chain.
// TIGHT: held to 80 SGPRs (8 waves per SIMD = four workgroups per CU; 21 scalars live in VGPR lanes around the prologue and the exit) -- for
// launches that put more than three workgroups on a CU; the loose variant (101 SGPRs: three workgroups per CU) is 1.5 us faster per
// launch when the group is a handful of narrow tables and every workgroup has a CU to itself.
template <int LR, bool DBG = false, bool TIGHT = false>
__global__ __launch_bounds__(512, TIGHT ? (LR == 3 ? 6 : 8) : 2) void slot_group(SlotGroupArgs args) {
	const SlotGroupWho who = slot_group_who(args);
	if (who.none) return;
	const uint32_t warm = slot_warm_next(args.entry[who.table]);
	const SlotBatchEntry e = slot_scalar_copy(args.entry[who.table]);
	const SlotRun& run = e.run;
	if (who.w >= (1u << (run.g - run.half)) || threadIdx.x >= run.threads) return;
	DevProblem P = slot_entry_problem(e, false);
	if (DBG) P.dbg_flags = e.pad2;   // timing experiments (WHAMD_SLOT_SKIP: results invalid)
	if ((LR == 2 || LR == 3) && (run.yflags & 1u)) {
		if (run.spec_id) slot_run_body<(LR == 3 ? 3 : 2), DBG, true, true>(P, run, e.prev, e.cur, who.w, e.score_out);
		else slot_run_body<(LR == 3 ? 3 : 2), DBG, false, true>(P, run, e.prev, e.cur, who.w, e.score_out);
	} else if (run.spec_id) slot_run_body<LR, DBG, true>(P, run, e.prev, e.cur, who.w, e.score_out);
	else slot_run_body<LR, DBG, false>(P, run, e.prev, e.cur, who.w, e.score_out);
	slot_warm_done(warm);
}
