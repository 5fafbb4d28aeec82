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
		if (!(DBG && (P.dbg_flags & 128u))) {   // (WHAMD_SLOT_SKIP 32 / 64 / 128: prologue loads switched off -- lane sums / entering cells / A, hot lines)
			a_g = P.slot_tab[run.tab_g + w * ncols + cl];
			a_w = P.slot_tab[run.tab_w + wave * ncols + cl];
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
	bool flip = false;
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
	// ---- exit: scatter into the next step's order (cells whose free-slot bits are zero hold the representatives)
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
	if (score_out && w == 0 && tid == 0) *score_out = D[0];
	if (DBG && P.dbg && w == 0 && tid == 0) {
		unsigned long long* d = P.dbg + (size_t)run.pad * 48;
		d[0] = t_issued - t_start; d[1] = t_loaded - t_start; d[2] = t_loop - t_loaded; d[3] = __builtin_readcyclecounter() - t_loop; d[4] = ncols; d[5] = 1;
	}
}

// The per-run tables of the prologue (SlotRun::tab_g / tab_w / tab_sl), once per table at create time: blockIdx.y = run.
__global__ __launch_bounds__(256) void slot_tables(DevProblem P, const SlotRun* __restrict__ runs, uint32_t* __restrict__ tab) {
	const SlotRun& run = runs[blockIdx.y];
	const uint32_t ncols = run.ncols, L = run.L, lr = run.lr, nwg = 1u << (run.g - run.half), nwaves = run.threads >> 6;
	const uint32_t n_g = nwg * ncols, n_w = nwaves * ncols, n_sl = ncols * 64u;
	const SlotRow* __restrict__ rows = P.slot_rows + run.row_off;
	const uint32_t ysh = run.yflags & 1u, ybias = ysh ? SLOT_YBIAS : 0u;   // Y form: X0 = 2 A + bias = (2 G + bias) + 2 W + 2 SL
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_g + n_w + n_sl; i += gridDim.x * blockDim.x) {
		if (i < n_g) {
			const uint32_t w = i / ncols, c = i % ncols;
			const SlotRow& row = rows[c];
			uint32_t acc = row.Cp;
			for (uint32_t s = L; s < L + run.g; ++s) acc += (uint32_t)row.dslot[s] & (0u - ((w >> (s - L)) & 1u));
			tab[run.tab_g + i] = (acc << ysh) + ybias;
		} else if (i < n_g + n_w) {
			const uint32_t q = i - n_g, wave = q / ncols, c = q % ncols;
			const SlotRow& row = rows[c];
			uint32_t acc = 0;
			for (uint32_t s = lr + 6u; s < L; ++s) acc += (uint32_t)row.dslot[s] & (0u - ((wave >> (s - lr - 6u)) & 1u));
			tab[run.tab_w + q] = acc << ysh;
		} else {
			const uint32_t q = i - n_g - n_w, c = q >> 6, lane = q & 63u;
			const SlotRow& row = rows[c];
			uint32_t acc = 0;
			for (uint32_t j = 0; j < (uint32_t)SLOT_LANE; ++j) acc += (uint32_t)row.dslot[lr + j] & (0u - ((lane >> j) & 1u));   // (= dlane[j]; a Y-form row reuses those words)
			tab[run.tab_sl + q] = acc << ysh;
		}
	}
}

// DBG: timing experiments (WHAMD_SLOT_SKIP switches parts off -- results invalid) and in-kernel cycle stamps (WHAMD_SLOT_STAMPS);
// the production instantiation carries none of it (a single wave issues one VALU instruction per ~8 cycles -- measured,
// scripts/micro/issue_rate.hip -- so every instruction on the column chain counts).
// SPEC: the run ends a backtrace chunk and leaves the seed of the speculative walk (instantiated separately: the other runs
// do not even carry the test).
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
	const SlotBatchEntry e = slot_scalar_copy(args.entry[blockIdx.y]);
	const SlotRun& run = e.run;
	if (blockIdx.x >= (1u << (run.g - run.half)) || threadIdx.x >= run.threads) return;
	DevProblem P = slot_entry_problem(e, false);
	if (DBG) P.dbg_flags = e.pad2;   // timing experiments (WHAMD_SLOT_SKIP: results invalid)
	if ((LR == 2 || LR == 3) && (run.yflags & 1u)) {
		if (run.spec_id) slot_run_body<(LR == 3 ? 3 : 2), DBG, true, true>(P, run, e.prev, e.cur, blockIdx.x, e.score_out);
		else slot_run_body<(LR == 3 ? 3 : 2), DBG, false, true>(P, run, e.prev, e.cur, blockIdx.x, e.score_out);
	} else if (run.spec_id) slot_run_body<LR, DBG, true>(P, run, e.prev, e.cur, blockIdx.x, e.score_out);
	else slot_run_body<LR, DBG, false>(P, run, e.prev, e.cur, blockIdx.x, e.score_out);
}
