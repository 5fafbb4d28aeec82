// dp_device.hip -- gfx950 kernels and the device driver of the wMEC / PedMEC forward pass + backtrace.
//
// What is computed (bit-exact restatement of src/pedigreedptable.cpp:177-335):
//   D_c[x][i]  = cost_{c,i}(x) (+) min_j ( Pr_{c-1}[x & lowmask_b][j] + popcount(i^j) * recomb_c ),  lowest j on ties
//   Pr_c[y][i] = min { D_c[x][i] : pext(x, fwd_mask_c) == y },  argmin = the x with the smallest Gray-code rank
// The reference walks x in reflected-Gray-code order with strict '<' updates; here every cell is evaluated
// independently (closed-form cost, no Gray stepping) and ties are broken with the key (value, gray_rank(x)).
//
// Path "column" (this file): one launch per column.
//   mode 0  column_step_fused : thread = one projection entry y; it enumerates the <= 2^4 cells that project onto y,
//                               writes Pr_c[y][*] coalesced and the winning ending-bit pattern / transmission argmin
//                               as ballot-packed bit planes (k-f+2*trios bits per entry instead of the reference's 8 bytes)
//   mode 1  column_step_keys  : many ending reads, tiny columns, or the last column: thread = (y, chunk of ending-bit
//                               patterns), 64-bit atomicMin on (value << 32 | rank << 4 | argj); column_finalize
//                               turns keys into Pr_c and a raw u32 backtrace record.
//   backtrace_kernel          : follows the stored argmins from the last column to the first (src/pedigreedptable.cpp:137-173).
// No MFMA (integer min-plus), no CUDA compatibility layer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "device_table.h"

namespace whamd {

#define HIP_TRY(expr)                                                                                 \
	do {                                                                                              \
		hipError_t err_ = (expr);                                                                     \
		if (err_ != hipSuccess) {                                                                     \
			msg = std::string(#expr) + " failed: " + hipGetErrorString(err_);                         \
			return WHAMD_ERR_DEVICE;                                                                  \
		}                                                                                             \
	} while (0)

namespace {

constexpr int QMAX = 4;  // a thread enumerates at most 2^QMAX ending-bit patterns itself

// Bit deposit through host-precomputed runs of the mask (gfx950 has no PDEP): the segment list is wave-uniform.
__device__ __forceinline__ uint32_t deposit(uint32_t v, const uint32_t* __restrict__ segs, uint32_t nseg) {
	uint32_t x = 0;
	for (uint32_t i = 0; i < nseg; ++i) {
		const uint32_t sg = segs[i];
		const uint32_t src = sg & 31u, dst = (sg >> 8) & 31u, len = (sg >> 16) & 31u;
		x |= ((v >> src) & ((1u << len) - 1u)) << dst;
	}
	return x;
}

// Position of x in the reflected Gray code sequence g(r) = r ^ (r >> 1)  (src/graycodes.cpp:26-43 visits g(0), g(1), ...).
__device__ __forceinline__ uint32_t gray_rank(uint32_t x) {
	x ^= x >> 1;
	x ^= x >> 2;
	x ^= x >> 4;
	x ^= x >> 8;
	x ^= x >> 16;
	return x;
}

// Per-launch staging of a column's cost data in LDS: 5-bit lookup tables of the per-individual sums L_s(x) and the
// term lists of every transmission value, so that a cell costs ceil(k/5) LDS reads per individual instead of a
// k-step loop over scalar global loads.
constexpr int COL_CHUNKS = 5, COL_MAXTERMS = 1024;
template <int T, int NIND>
struct ColumnStage {
	int32_t lut[NIND][COL_CHUNKS][32];
	DevTerm terms[COL_MAXTERMS];
	uint32_t tptr[T + 1];
};

template <int T, int NIND>
__device__ __forceinline__ void stage_column(ColumnStage<T, NIND>& S, const DevProblem& P, const DevColumn& col) {
	const int32_t* __restrict__ dl = P.delta + col.delta_off;
	const uint32_t k = col.k;
	for (uint32_t idx = threadIdx.x; idx < (uint32_t)(NIND * COL_CHUNKS * 32); idx += blockDim.x) {
		const uint32_t s = idx / (COL_CHUNKS * 32), chunk = (idx / 32) % COL_CHUNKS, v = idx & 31u;
		int32_t sum = 0;
#pragma unroll
		for (int j = 0; j < 5; ++j) {
			const uint32_t bit = chunk * 5 + j;
			if (bit < k && ((v >> j) & 1u)) sum += dl[s * k + bit];
		}
		S.lut[s][chunk][v] = sum;
	}
	const uint32_t* __restrict__ tp = P.term_ptr + col.term_off;
	const uint32_t t0 = tp[0], nterms = min(tp[T] - t0, (uint32_t)COL_MAXTERMS);
	for (uint32_t i = threadIdx.x; i < nterms; i += blockDim.x) S.terms[i] = P.terms[t0 + i];
	if (threadIdx.x <= (uint32_t)T) S.tptr[threadIdx.x] = min(tp[threadIdx.x] - t0, (uint32_t)COL_MAXTERMS);
	__syncthreads();
}

// cost_{c,t}(x) for all t (get_cost, src/pedigreecolumncostcomputer.cpp:101-114) from the per-individual sums L_s(x).
template <int T, int NIND>
__device__ __forceinline__ void cell_costs(uint32_t x, const ColumnStage<T, NIND>& S, uint32_t nchunks, uint32_t (&cost)[T]) {
	int32_t L[NIND];
#pragma unroll
	for (int s = 0; s < NIND; ++s) L[s] = 0;
	for (uint32_t c = 0; c < nchunks; ++c) {
		const uint32_t v = (x >> (5 * c)) & 31u;
#pragma unroll
		for (int s = 0; s < NIND; ++s) L[s] += S.lut[s][c][v];
	}
#pragma unroll
	for (int t = 0; t < T; ++t) {
		uint32_t best = 0xFFFFFFFFu;
		const uint32_t e = S.tptr[t + 1];
		for (uint32_t q = S.tptr[t]; q < e; ++q) {
			const DevTerm tm = S.terms[q];
			uint32_t v = tm.c;
#pragma unroll
			for (int s = 0; s < NIND; ++s) {
				v += ((tm.plus >> s) & 1u) ? (uint32_t)L[s] : 0u;
				v -= ((tm.minus >> s) & 1u) ? (uint32_t)L[s] : 0u;
			}
			best = min(best, v);
		}
		cost[t] = best;
	}
}

// D[i] and argj[i] of one cell (src/pedigreedptable.cpp:264-300).  prev == nullptr for column 0.
template <int T>
__device__ __forceinline__ void cell_dp(const uint32_t (&cost)[T], const uint32_t* __restrict__ prev, uint32_t z,
                                        uint32_t recomb, uint32_t (&D)[T], uint32_t (&aj)[T]) {
	uint32_t pv[T];
	if (prev) {
		if constexpr (T == 1) {
			pv[0] = prev[z];
		} else {
			const uint4* p4 = reinterpret_cast<const uint4*>(prev + (size_t)z * T);
#pragma unroll
			for (int q = 0; q < T / 4; ++q) {
				const uint4 v = p4[q];
				pv[4 * q] = v.x; pv[4 * q + 1] = v.y; pv[4 * q + 2] = v.z; pv[4 * q + 3] = v.w;
			}
		}
	} else {
#pragma unroll
		for (int j = 0; j < T; ++j) pv[j] = 0;
	}
#pragma unroll
	for (int i = 0; i < T; ++i) {
		uint32_t m = 0xFFFFFFFFu, mj = 0;
		if (cost[i] != 0xFFFFFFFFu) {
#pragma unroll
			for (int j = 0; j < T; ++j) {
				if (pv[j] != 0xFFFFFFFFu) {
					const uint32_t val = cost[i] + pv[j] + (uint32_t)__popc((unsigned)(i ^ j)) * recomb;
					if (val < m) { m = val; mj = j; }
				}
			}
		}
		D[i] = m;
		aj[i] = mj;
	}
}

template <int T, int NIND>
__global__ __launch_bounds__(256) void column_step_fused(DevProblem P, uint32_t c, const uint32_t* __restrict__ prev,
                                                          uint32_t* __restrict__ cur) {
	const DevColumn col = P.cols[c];
	__shared__ ColumnStage<T, NIND> stage;
	stage_column<T, NIND>(stage, P, col);
	const uint32_t nchunks = (col.k + 4) / 5;
	const uint32_t y = blockIdx.x * blockDim.x + threadIdx.x;  // grid covers exactly 2^f entries (f >= 6)
	const uint32_t* __restrict__ segs = P.segs + col.seg_off;
	const uint32_t xbase = deposit(y, segs, col.nseg_fwd);
	const uint32_t lowmask = (1u << col.b) - 1u;
	const uint32_t* pr = c ? prev : nullptr;
	uint32_t bD[T], bR[T], bV[T];
#pragma unroll
	for (int i = 0; i < T; ++i) { bD[i] = 0xFFFFFFFFu; bR[i] = 0xFFFFFFFFu; bV[i] = 0; }
	const uint32_t ne = 1u << col.ebits;
	for (uint32_t e = 0; e < ne; ++e) {
		const uint32_t x = xbase | deposit(e, segs + col.nseg_fwd, col.nseg_end);
		uint32_t cost[T], D[T], aj[T];
		cell_costs<T, NIND>(x, stage, nchunks, cost);
		cell_dp<T>(cost, pr, x & lowmask, col.recomb, D, aj);
		const uint32_t r = gray_rank(x);
#pragma unroll
		for (int i = 0; i < T; ++i) {
			const bool better = (D[i] < bD[i]) || (D[i] == bD[i] && r < bR[i]);
			if (better) { bD[i] = D[i]; bR[i] = r; bV[i] = e | (aj[i] << col.ebits); }
		}
	}
	if constexpr (T == 1) {
		cur[y] = bD[0];
	} else {
		uint4* c4 = reinterpret_cast<uint4*>(cur + (size_t)y * T);
#pragma unroll
		for (int q = 0; q < T / 4; ++q) c4[q] = make_uint4(bD[4 * q], bD[4 * q + 1], bD[4 * q + 2], bD[4 * q + 3]);
	}
	// backtrace record: nplanes bit planes per transmission value, one ballot word per 64 consecutive y
	unsigned long long* planes = reinterpret_cast<unsigned long long*>(P.bt + col.bt_off);
	const uint32_t words = 1u << (col.f - 6);
	const uint32_t w = y >> 6;
	for (uint32_t p = 0; p < col.nplanes; ++p) {
#pragma unroll
		for (int i = 0; i < T; ++i) {
			const unsigned long long word = __ballot((bV[i] >> p) & 1u);
			if ((threadIdx.x & 63u) == 0) planes[(size_t)(p * T + i) * words + w] = word;
		}
	}
}

template <int T, int NIND>
__global__ __launch_bounds__(256) void column_step_keys(DevProblem P, uint32_t c, const uint32_t* __restrict__ prev,
                                                         uint32_t total_threads) {
	const DevColumn col = P.cols[c];
	__shared__ ColumnStage<T, NIND> stage;
	stage_column<T, NIND>(stage, P, col);
	const uint32_t nchunks = (col.k + 4) / 5;
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= total_threads) return;
	const uint32_t y = gid & ((1u << col.f) - 1u);
	const uint32_t chunk = gid >> col.f;
	const uint32_t* __restrict__ segs = P.segs + col.seg_off;
	const uint32_t xbase = deposit(y, segs, col.nseg_fwd);
	const uint32_t lowmask = (1u << col.b) - 1u;
	const uint32_t* pr = c ? prev : nullptr;
	unsigned long long best[T];
#pragma unroll
	for (int i = 0; i < T; ++i) best[i] = ~0ull;
	const uint32_t ne = 1u << col.eloop;
	for (uint32_t el = 0; el < ne; ++el) {
		const uint32_t e = (chunk << col.eloop) | el;
		const uint32_t x = xbase | deposit(e, segs + col.nseg_fwd, col.nseg_end);
		uint32_t cost[T], D[T], aj[T];
		cell_costs<T, NIND>(x, stage, nchunks, cost);
		cell_dp<T>(cost, pr, x & lowmask, col.recomb, D, aj);
		const uint32_t r = gray_rank(x);
#pragma unroll
		for (int i = 0; i < T; ++i) {
			const unsigned long long key = ((unsigned long long)D[i] << 32) | ((unsigned long long)r << 4) | aj[i];
			best[i] = min(best[i], key);
		}
	}
	// Lanes whose indices differ by a multiple of 2^f hold candidates for the SAME projection entry (the last column has
	// f = 0: a million atomics on one word took 0.76 ms): reduce them inside the wave first, one atomic per entry and wave.
	const bool wave_reduce = col.f < 6u && total_threads >= 64u;  // total_threads is a power of two: every wave is full
	if (wave_reduce) {
		for (uint32_t stride = 32; stride >= (1u << col.f); stride >>= 1) {
#pragma unroll
			for (int i = 0; i < T; ++i) best[i] = min(best[i], (unsigned long long)__shfl_xor(best[i], (int)stride));
			if (stride == 1u) break;
		}
		if ((threadIdx.x & 63u) >> col.f) return;
	}
#pragma unroll
	for (int i = 0; i < T; ++i) atomicMin(&P.keys[(size_t)y * T + i], best[i]);
}

// keys -> Pr_c (value) + raw u32 backtrace record (rank << 4 | argj); re-arms the key scratch.
__global__ __launch_bounds__(256) void column_finalize(DevProblem P, uint32_t c, uint32_t* __restrict__ cur, uint32_t entries) {
	const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= entries) return;
	const DevColumn col = P.cols[c];
	const unsigned long long key = P.keys[idx];
	P.keys[idx] = ~0ull;
	cur[idx] = (uint32_t)(key >> 32);
	reinterpret_cast<uint32_t*>(P.bt + col.bt_off)[idx] = (uint32_t)key;
	if (col.is_last) P.last_keys[idx] = key;
}

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
template <uint32_t MODE, int NC>
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
		res_finish_entries<MODE, NC>(acc, base, mL0, PG, pbits, bufQ, rec, t);
	}
}

// Packed 16-bit evaluation (resident.h pk_ok): two cells per instruction, the costs of the folded columns are added in
// 16 bits and widened once.  Reads only Q0..Q3 of this column and Q0, Q1 of every folded one.
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_pk(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

template <uint32_t MODE>
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
		res_finish_entries<MODE, NC>(acc, base, mL0, PG, q3.w, bufQ, rec, t);
	}
}

// PMC finding (profiles/r01_pmc_resident_v1.txt): the per-column loop is bound by instruction ISSUE, first of all by the
// scalar unit the 16 waves of a workgroup share -- so the loop keeps per-column constants in vector registers (LDS
// broadcast reads), lets whole waves without work branch straight to the barrier, and records the argmin bits as one
// byte per thread (no ballot / exec-mask sequences).
template <bool DBG>
__global__ __launch_bounds__(1024) void resident_segment(DevProblem P, ResSegment sg, const uint32_t* __restrict__ prev,
                                                          uint32_t* __restrict__ cur) {
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	const unsigned long long t_begin = DBG ? __builtin_readcyclecounter() : 0ull;
	touch_kernel_arguments<sizeof(DevProblem) + sizeof(ResSegment) + 16>();
	const uint32_t w = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
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
		uint4 vd[2], vt[4];
		uint32_t vs[4];
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + tid; vd[u] = i < ndesc ? desc_at(i) : make_uint4(0, 0, 0, 0); }
#pragma unroll
		for (int u = 0; u < 4; ++u) { const uint32_t i = u * NT + tid; vt[u] = i < ntab ? gt[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t l = u * NT + tid;
			vs[u] = l < nslice ? prev[wpart | deposit_args(l, sg.in_local, sg.n_in_local)] : 0u;
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
		for (uint32_t l = 4 * NT + tid; l < nslice; l += NT) bufP[l] = prev[wpart | deposit_args(l, sg.in_local, sg.n_in_local)];
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
					if (mode == RES_MODE_E0) res_pk_column<RES_MODE_E0>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
					else if (mode == RES_MODE_E1_HIGH) res_pk_column<RES_MODE_E1_HIGH>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
					else if (mode == RES_MODE_E1_BIT0) res_pk_column<RES_MODE_E1_BIT0>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
					else res_pk_column<RES_MODE_E1_BIT1>(ldsc, tab, ci, nfold, bufP, bufQ, stage, tid, NT, nthr, q0, q2);
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
	for (uint32_t l = tid; l < (1u << sg.Lf_last); l += NT) cur[wout | deposit_args(l, sg.out_local, sg.n_out_local)] = bufP[l];
	unsigned long long* rec = reinterpret_cast<unsigned long long*>(P.bt + (((unsigned long long)sg.bt_hi << 32) | sg.bt_lo)) + (size_t)w * sg.stage_words;
	const unsigned long long* st64 = reinterpret_cast<const unsigned long long*>(stage);
	if (!(DBG && (P.dbg_flags & 2u)))
	for (uint32_t i = tid; i < sg.stage_words; i += NT) rec[i] = st64[i];
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
template <bool DBG>
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
	for (uint32_t l = tid; l < (1u << sg.Lf_last); l += NT) c4[wout | deposit_args(l, sg.out_local, sg.n_out_local)] = bufP[l];
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

// Backtrace (src/pedigreedptable.cpp:137-173); out: index / transmission per column, out_score[0] = optimum.
// The steps of the forward plan are walked in reverse (`units`, newest first).  For a resident run the argmin bits the
// path can touch all belong to ONE workgroup's record (the grid-read bits of the path do not change inside a run), so
// the workgroup copies that record (a few KiB) into LDS with one coalesced load while it prefetches the NEXT run's
// column records and the header of the run after that; one wave then follows the path with LDS latency instead of one
// dependent HBM access per column.
//
// `with_last_column` != 0: units[0] is the table's last column (its optimum comes from P.last_keys), the walk starts at
// units[1].  == 0: the units are the runs of ONE connected component that ends before the table does (its last column
// projects onto a single entry): the walk starts at units[0] with entry 0 and no score is written -- several such
// launches run side by side on different streams.
__global__ __launch_bounds__(1024) void backtrace_kernel(DevProblem P, const BtUnit* __restrict__ units, uint32_t n_units,
                                                         uint32_t with_last_column, uint32_t* __restrict__ path_index,
                                                         uint32_t* __restrict__ path_trans, uint32_t* __restrict__ out_score) {
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	uint32_t* recs0 = smem;                                   // 2 x RES_MAXCOLS * 32 words: column records (double buffer)
	uint32_t* hdr = smem + 2 * RES_MAXCOLS * 32;              // 4 x 32 words: unit headers (ring)
	uint32_t* xshare = hdr + 128;                             // 4 words
	uint32_t* cells = xshare + 4;                             // RES_MAXCOLS words: local cell index of the path per column
	uint32_t* tsarr = cells + RES_MAXCOLS;                    // RES_MAXCOLS words: transmission value of the path per column
	unsigned long long* stage = reinterpret_cast<unsigned long long*>(tsarr + RES_MAXCOLS);
	const uint32_t lane = threadIdx.x, NT = blockDim.x;
	const uint32_t n = P.n_cols, T = P.T;
	const uint32_t u_first = with_last_column ? 1u : 0u;
	uint32_t x = 0, tprev = 0;
	if (with_last_column) {
		// optimum of the last column: first (rank(x), i) attaining the minimum (strict '<' scan, :306-315)
		unsigned long long bestk = ~0ull;
		uint32_t t = 0;
		for (uint32_t i = 0; i < T; ++i) {
			const unsigned long long key = P.last_keys[i];
			if ((key >> 4) < (bestk >> 4)) { bestk = key; t = i; }
		}
		if (bestk == ~0ull) {  // unreachable for valid inputs (the host rejects Mendelian conflicts); keep defined output
			if (lane == 0) out_score[0] = 0xFFFFFFFFu;
			bestk = 0;
		} else if (lane == 0) {
			out_score[0] = (uint32_t)(bestk >> 32);
		}
		const uint32_t rlast = (uint32_t)(bestk >> 4) & 0x0FFFFFFFu;
		x = rlast ^ (rlast >> 1);
		tprev = (uint32_t)bestk & 15u;
		if (lane == 0) {
			path_index[n - 1] = x;
			path_trans[n - 1] = t;
		}
	}
	// every unit from u_first on yields x_c from x_{c+1}.
	// prime the pipeline: headers of the first two units, records of the first
	if (lane < 64) {
		const uint32_t u = u_first + (lane >> 5);
		if (u < n_units) hdr[(u & 3u) * 32 + (lane & 31u)] = reinterpret_cast<const uint32_t*>(units + u)[lane & 31u];
	}
	__syncthreads();
	if (n_units > u_first && hdr[(u_first & 3u) * 32] == 1u) {
		const uint32_t* h1 = hdr + (u_first & 3u) * 32;
		const uint32_t* __restrict__ g1 = reinterpret_cast<const uint32_t*>(P.res_bt + h1[3]);
		for (uint32_t i = lane; i < h1[2] * 32; i += NT) recs0[(u_first & 1u) * RES_MAXCOLS * 32 + i] = g1[i];
	}
	__syncthreads();
	unsigned long long bt_load = 0, bt_walk = 0, bt_runs = 0, bt_a = 0, bt_b = 0, bt_c = 0;
	for (uint32_t ui = u_first; ui < n_units; ++ui) {
		const unsigned long long tb0 = P.dbg ? __builtin_readcyclecounter() : 0ull;
		const uint32_t* h = hdr + (ui & 3u) * 32;
		const uint32_t kind = h[0], c0 = h[1], ncols = h[2];
		uint32_t* recs = recs0 + (ui & 1u) * RES_MAXCOLS * 32;
		// prefetch: header of unit ui + 2, records of unit ui + 1 (its header arrived one iteration ago)
		uint32_t hv = 0, wrun = 0;
		const bool hload = lane < 32 && ui + 2 < n_units;
		if (hload) hv = reinterpret_cast<const uint32_t*>(units + ui + 2)[lane];
		const uint32_t* hn = hdr + ((ui + 1) & 3u) * 32;
		const bool next_run = ui + 1 < n_units && hn[0] == 1u;
		const uint32_t nrec = next_run ? hn[2] * 32 : 0u;
		const uint32_t* __restrict__ gnext = reinterpret_cast<const uint32_t*>(P.res_bt + (next_run ? hn[3] : 0u));
		uint32_t rv[2];
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; rv[u] = i < nrec ? gnext[i] : 0u; }
		if (kind == 0) {
			// ---- one column through the column kernels' records (global loads; rare in steady state)
			const uint32_t c = c0;
			// header words of a column unit: 4 f, 5 mode, 6 nplanes, 7 ebits, 8/9 record offset, 10 nseg_fwd, 11 nseg_end,
			// 12..27 deposit runs (if word 28 is set; else they are read from the column descriptor)
			const uint32_t cf = h[4], cmode = h[5], cnplanes = h[6], cebits = h[7], nsf = h[10], nse = h[11];
			const unsigned long long cbt = ((unsigned long long)h[9] << 32) | h[8];
			const uint32_t* segs = h[28] ? (h + 12) : (P.segs + P.cols[c].seg_off);
			const uint32_t y = x & ((1u << cf) - 1u);
			uint32_t xp, aj;
			if (cmode == 0) {
				const unsigned long long* planes = reinterpret_cast<const unsigned long long*>(P.bt + cbt);
				const uint32_t words = 1u << (cf - 6);
				unsigned long long wv[8];
#pragma unroll
				for (int p = 0; p < 8; ++p) wv[p] = (uint32_t)p < cnplanes ? planes[(size_t)(p * T + tprev) * words + (y >> 6)] : 0ull;
				uint32_t v = 0;
#pragma unroll
				for (int p = 0; p < 8; ++p) v |= (uint32_t)((wv[p] >> (y & 63u)) & 1ull) << p;
				const uint32_t e = v & ((1u << cebits) - 1u);
				aj = v >> cebits;
				xp = deposit(y, segs, nsf) | deposit(e, segs + nsf, nse);
			} else {
				const uint32_t raw = reinterpret_cast<const uint32_t*>(P.bt + cbt)[(size_t)y * T + tprev];
				const uint32_t r = raw >> 4;
				xp = r ^ (r >> 1);
				aj = raw & 15u;
			}
			if (lane == 0) {
				path_index[c] = xp;
				path_trans[c] = tprev;
			}
			tprev = aj;
			x = xp;
		} else {
			// ---- resident run [c0, c0 + ncols): this workgroup's record -> LDS
			const uint32_t g = h[4], Lf_last = h[5], stage_words = h[6], n_wext = h[7];
			const uint32_t yexit = x & ((1u << (Lf_last + g)) - 1u);
			uint32_t w = 0;
			for (uint32_t i = 0; i < n_wext; ++i) {
				const uint32_t r = h[12 + i];
				w |= ((yexit >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
			}
			wrun = w;
			const unsigned long long* __restrict__ gst = reinterpret_cast<const unsigned long long*>(
				P.bt + (((unsigned long long)h[9] << 32) | h[8])) + (size_t)w * stage_words;
			unsigned long long sv[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; sv[u] = i < stage_words ? gst[i] : 0ull; }
#pragma unroll
			for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; if (i < stage_words) stage[i] = sv[u]; }
			for (uint32_t i = 2 * NT + lane; i < stage_words; i += NT) stage[i] = gst[i];
		}
		// land the prefetches
		if (hload) hdr[((ui + 2) & 3u) * 32 + lane] = hv;
		{
			uint32_t* rnext = recs0 + ((ui + 1) & 1u) * RES_MAXCOLS * 32;
#pragma unroll
			for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; if (i < nrec) rnext[i] = rv[u]; }
			for (uint32_t i = 2 * NT + lane; i < nrec; i += NT) rnext[i] = gnext[i];
		}
		__syncthreads();
		const unsigned long long tb1 = P.dbg ? __builtin_readcyclecounter() : 0ull;
		if (kind == 1) {
			if (lane < 64) {  // one wave follows the path; the others only helped with the copies
				// local exit index of the path
				const uint32_t yexit = x & ((1u << (h[5] + h[4])) - 1u);
				uint32_t l = 0;
				for (uint32_t i = 0; i < h[10]; ++i) {
					const uint32_t r = h[18 + i];
					l |= ((yexit >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
				}
				// sequential part, in local index space: only columns where a read ends touch the record.  Lanes keep the
				// per-column parameters in registers; the loop fetches them with v_readlane (off the dependent chain), so the
				// chain per visited column is: mask, record byte from LDS, bit insert.
				const unsigned long long tw0 = P.dbg ? __builtin_readcyclecounter() + (l & 0u) : 0ull;
				const uint32_t n_active = h[11] & 0xFFFFu, simple = h[11] >> 16;
				const uint32_t* rmine = recs + (lane < ncols ? lane : 0u) * 32;
				uint32_t tcur = tprev, mycell = 0, myts = 0;
				if (simple == 2u) {
					// trio, at most one read ends per column: every column reads one record byte (the transmission argmin lives
					// there), the chain per column is mask, byte, (bit insert); parameters by v_readlane from lane ci
					const uint4 q0 = *reinterpret_cast<const uint4*>(rmine);      // Lf, ebits, layout, stage_off
					const uint32_t p_mask = (1u << q0.x) - 1u, p_soff = q0.w * 8u, p_eb = q0.y | (rmine[5] << 8);
					const uint8_t* stage8 = reinterpret_cast<const uint8_t*>(stage);
					for (uint32_t ci = ncols; ci-- > 0;) {
						const uint32_t s_mask = __builtin_amdgcn_readlane(p_mask, ci), s_soff = __builtin_amdgcn_readlane(p_soff, ci),
						               s_eb = __builtin_amdgcn_readlane(p_eb, ci);
						const uint32_t lout = l & s_mask;
						const uint32_t fld = stage8[s_soff + lout * 4u + tcur];
						const uint32_t e0 = s_eb >> 8;
						const uint32_t with_bit = insert_zero(lout, e0) | ((fld & 1u) << e0);
						const uint32_t cell = (s_eb & 255u) ? with_bit : lout;
						if (lane == ci) { mycell = cell; myts = tcur; }
						tcur = (fld >> 3) & 3u;
						l = cell;
					}
				} else if (simple) {
					// single individual, every record one byte per thread: the chain visits only the columns in which a read ends
					// (ResBacktrace kpos / src / cmask / kcol); lane k holds the parameters of chain position k
					const uint32_t kc = rmine[31] < ncols ? rmine[31] : 0u;
					const uint32_t* rk = recs + kc * 32;
					const uint32_t c_cmask = rk[30], c_soff = rk[3] * 8u, c_e0 = rk[5];
					const uint32_t my_kpos = rmine[28], my_src = rmine[29], my_cmask = rmine[30];
					const uint8_t* stage8 = reinterpret_cast<const uint8_t*>(stage);
					const uint32_t l_exit = l;
					for (uint32_t k = 0; k < n_active; ++k) {
						const uint32_t s_cmask = __builtin_amdgcn_readlane(c_cmask, k), s_soff = __builtin_amdgcn_readlane(c_soff, k),
						               s_e0 = __builtin_amdgcn_readlane(c_e0, k);
						const uint32_t lout = l & s_cmask;
						const uint32_t byte = stage8[s_soff + (lout >> 2)];
						const uint32_t cell = insert_zero(lout, s_e0) | (((byte >> (lout & 3u)) & 1u) << s_e0);
						if (my_kpos == k) mycell = cell;
						l = cell;
					}
					// columns without an ending read: the cell of the next active column above (or the exit index), masked
					const uint32_t from = __shfl(mycell, my_src == RES_BT_NONE ? 0u : my_src);
					if (my_kpos == RES_BT_NONE) mycell = (my_src == RES_BT_NONE ? l_exit : from) & my_cmask;
				} else {
					const uint4 q0 = *reinterpret_cast<const uint4*>(rmine);      // Lf, ebits, layout, stage_off
					const uint4 q1 = *reinterpret_cast<const uint4*>(rmine + 4);  // nwords, epos0, epos1, epos2
					const uint32_t p_mask = (1u << q0.x) - 1u, p_eb = q0.y | (q0.z << 8), p_soff = q0.w * 8u, p_e0 = q1.y, p_e1 = q1.z, p_e2 = q1.w, p_nw = q1.x;
					const uint8_t* stage8 = reinterpret_cast<const uint8_t*>(stage);
					for (uint32_t ci = ncols; ci-- > 0;) {
						const uint32_t s_mask = __builtin_amdgcn_readlane(p_mask, ci), s_eb = __builtin_amdgcn_readlane(p_eb, ci);
						const uint32_t lout = l & s_mask;
						uint32_t cell = lout;
						const uint32_t eb = s_eb & 255u, layout = s_eb >> 8;
						if (layout == 2u) {  // trio: one byte per (entry, transmission value): ending-read bits | argj << 3
							const uint32_t s_soff = __builtin_amdgcn_readlane(p_soff, ci);
							const uint32_t fld = stage8[s_soff + lout * 4u + tcur] & 31u;
							if (eb) {
								const uint32_t epos[3] = {(uint32_t)__builtin_amdgcn_readlane(p_e0, ci), (uint32_t)__builtin_amdgcn_readlane(p_e1, ci),
								                          (uint32_t)__builtin_amdgcn_readlane(p_e2, ci)};
								uint32_t bits = 0;
	#pragma unroll
								for (int q = 0; q < 3; ++q) {
									if ((uint32_t)q < eb) {
										cell = insert_zero(cell, epos[q]);
										bits |= ((fld >> q) & 1u) << epos[q];
									}
								}
								cell |= bits;
							}
							if (lane == ci) myts = tcur;
							tcur = fld >> 3;
						} else if (eb) {
							const uint32_t s_soff = __builtin_amdgcn_readlane(p_soff, ci), s_e0 = __builtin_amdgcn_readlane(p_e0, ci);
							if (layout == 1u) {  // one byte per thread: bit (lout & 3) of byte lout >> 2
								const uint32_t byte = stage8[s_soff + (lout >> 2)];
								cell = insert_zero(lout, s_e0) | (((byte >> (lout & 3u)) & 1u) << s_e0);
							} else {             // ballot planes, up to 3 ending reads (ascending positions)
								const uint32_t epos[3] = {s_e0, (uint32_t)__builtin_amdgcn_readlane(p_e1, ci), (uint32_t)__builtin_amdgcn_readlane(p_e2, ci)};
								const uint32_t s_nw = __builtin_amdgcn_readlane(p_nw, ci);
								uint32_t bits = 0;
	#pragma unroll
								for (int q = 0; q < 3; ++q) {
									if ((uint32_t)q < eb) {
										cell = insert_zero(cell, epos[q]);
										const unsigned long long word = stage[(s_soff >> 3) + q * s_nw + (lout >> 6)];
										bits |= (uint32_t)((word >> (lout & 63u)) & 1ull) << epos[q];
									}
								}
								cell |= bits;
							}
						}
						if (lane == ci) mycell = cell;
						l = cell;
					}
				}
				const unsigned long long tw1 = P.dbg ? __builtin_readcyclecounter() + (l & 0u) : 0ull;
				// logical indices, one lane per column
				uint32_t xl = 0;
				if (lane < ncols) {
					const uint32_t* rb = recs + lane * 32;
					const uint32_t cell = mycell;
					const uint32_t ng = rb[8], nl = rb[9];
					for (uint32_t i = 0; i < ng; ++i) {
						const uint32_t r = rb[10 + i];
						xl |= ((wrun >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
					}
					for (uint32_t i = 0; i < nl; ++i) {
						const uint32_t r = rb[18 + i];
						xl |= ((cell >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
					}
					path_index[c0 + lane] = xl;
					path_trans[c0 + lane] = rb[2] == 2u ? myts : 0u;
				}
				if (lane == 0) { xshare[0] = xl; xshare[1] = tcur; }
				if (P.dbg) { const unsigned long long tw2 = __builtin_readcyclecounter() + (xl & 0u); bt_a += tw0 - tb1; bt_b += tw1 - tw0; bt_c += tw2 - tw1; }
			}
			__syncthreads();
			x = xshare[0];
			tprev = xshare[1];
			if (P.dbg) { bt_load += tb1 - tb0; bt_walk += __builtin_readcyclecounter() - tb1; bt_runs++; }
		}
	}
	if (P.dbg && lane == 0) {
		unsigned long long* d = P.dbg + P.dbg_wg_off + 4 * 512 * 2;
		d[0] = bt_load; d[1] = bt_walk; d[2] = bt_runs; d[3] = bt_a; d[4] = bt_b; d[5] = bt_c;
	}
}

// ---------------------------------------------------------------------------------------------- launch tables
using FusedFn = void (*)(DevProblem, uint32_t, const uint32_t*, uint32_t*);
using KeysFn = void (*)(DevProblem, uint32_t, const uint32_t*, uint32_t);

template <int T, int NIND>
void pick(FusedFn& ff, KeysFn& kf) {
	ff = column_step_fused<T, NIND>;
	kf = column_step_keys<T, NIND>;
}

bool select_kernels(uint32_t T, uint32_t n_ind, FusedFn& ff, KeysFn& kf) {
	const uint32_t ni = n_ind ? n_ind : 1;  // an empty pedigree has no terms to add; NIND=1 with zero deltas is equivalent
	ff = nullptr;
	kf = nullptr;
#define WHAMD_CASE(TT, NN) if (T == TT && ni == NN) { pick<TT, NN>(ff, kf); return true; }
	WHAMD_CASE(1, 1) WHAMD_CASE(1, 2) WHAMD_CASE(1, 3) WHAMD_CASE(1, 4) WHAMD_CASE(1, 5) WHAMD_CASE(1, 6)
	WHAMD_CASE(4, 3) WHAMD_CASE(4, 4) WHAMD_CASE(4, 5) WHAMD_CASE(4, 6)
	WHAMD_CASE(16, 4) WHAMD_CASE(16, 5) WHAMD_CASE(16, 6)
#undef WHAMD_CASE
	return false;
}

}  // namespace

// ================================================================================================ DeviceTable

struct DeviceTable::Impl {
	int device = 0;
	hipStream_t stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
	std::vector<void*> allocations;
	DevColumn* d_cols = nullptr;
	uint32_t* d_pr[2] = {nullptr, nullptr};
	uint32_t* d_path_index = nullptr;
	uint32_t* d_path_trans = nullptr;
	uint32_t* d_score = nullptr;
	BtUnit* d_units = nullptr;
	std::vector<BtUnit> units;
	size_t bt_lds = 0;
	std::vector<DevColumn> cols;
	ResidentPlan plan;
	DevProblem dp{};
	FusedFn fused = nullptr;
	KeysFn keysfn = nullptr;
	size_t key_entries = 0;
	std::string path = "auto";
	int l_pref = 11;
	bool fold = true;
	uint64_t bt_bytes = 0;
	uint64_t launches = 0;
	bool enqueue_open = false;  // resumable enqueue (enqueue_some)
	uint32_t* h_pinned = nullptr;  // [2 n + 1 + jobs]: path index, path transmission, score of the final job, scores of the others
	// A job is a sequence of forward steps followed by ONE backtrace launch.  Job 0 ("final") is the last connected
	// component (it ends with the table's last column, whose optimum comes from the key scratch); every other job is
	// one earlier connected component: it starts from cost 0, its last column projects onto a single entry -- that value
	// is added on the host -- and its backtrace starts at entry 0.  Jobs are independent, so they are spread over lanes (streams) and overlap.
	struct Job {
		std::vector<uint32_t> steps;  // indices into plan.steps, execution order
		uint32_t unit_off = 0, unit_count = 0;
		bool final = false;
	};
	struct Lane {
		hipStream_t stream = nullptr;  // lane 0: the table's stream
		hipEvent_t done = nullptr;
		uint32_t* d_pr[2] = {nullptr, nullptr};
		unsigned long long* d_keys = nullptr;  // atomic-min scratch of the per-column kernels (one per lane: components overlap)
		std::vector<uint32_t> jobs;
		size_t job_i = 0, step_i = 0;  // cursor of the resumable enqueue
		uint32_t flip = 0;             // every step reads d_pr[flip] and writes d_pr[flip ^ 1]
	};
	std::vector<Job> jobs;
	std::vector<Lane> lanes;
	uint32_t* d_job_scores = nullptr;
	hipEvent_t ev_ready = nullptr;
	int max_lanes = 4;  // measured: 4 streams saturate the dispatch rate (~200 k launches/s); more streams (or more hardware queues) lose
	size_t lanes_open = 0;

	void launch_step(const Problem& p, const Step& step, const Lane& lane, const uint32_t* prev, uint32_t* cur, uint64_t& launches);
	whamd_status_t submit_lane(const Problem& p, size_t li, uint64_t max_launches, uint64_t& launches, bool& finished, std::string& msg);

	void release_lanes() {
		for (size_t i = 1; i < lanes.size(); ++i) {
			if (lanes[i].done) (void)hipEventDestroy(lanes[i].done);
			if (lanes[i].stream) (void)hipStreamDestroy(lanes[i].stream);
		}
		lanes.clear();
		jobs.clear();
	}

	void release() {
		release_lanes();
		for (void* a : allocations) (void)hipFree(a);
		allocations.clear();
		if (h_pinned) (void)hipHostFree(h_pinned);
		h_pinned = nullptr;
		d_cols = nullptr;
		d_units = nullptr;
		d_pr[0] = d_pr[1] = nullptr;
		d_path_index = d_path_trans = d_score = nullptr;
	}
};

DeviceTable::DeviceTable() : impl_(new Impl()) {}

DeviceTable::~DeviceTable() {
	if (impl_) {
		(void)hipSetDevice(impl_->device);
		impl_->release();
		if (impl_->ev0) (void)hipEventDestroy(impl_->ev0);
		if (impl_->ev1) (void)hipEventDestroy(impl_->ev1);
		if (impl_->ev2) (void)hipEventDestroy(impl_->ev2);
		if (impl_->ev3) (void)hipEventDestroy(impl_->ev3);
		if (impl_->ev_ready) (void)hipEventDestroy(impl_->ev_ready);
		if (impl_->stream) (void)hipStreamDestroy(impl_->stream);
		delete impl_;
	}
}

void DeviceTable::release_device() {
	Impl& m = *impl_;
	(void)hipSetDevice(m.device);
	m.release();
	for (hipEvent_t* e : {&m.ev0, &m.ev1, &m.ev2, &m.ev3, &m.ev_ready}) {
		if (*e) (void)hipEventDestroy(*e);
		*e = nullptr;
	}
	if (m.stream) (void)hipStreamDestroy(m.stream);
	m.stream = nullptr;
}

int DeviceTable::device_count() {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// Runs of set bits of `mask` as deposit segments (compact position | mask position << 8 | length << 16).
static void append_segments(uint32_t mask, std::vector<uint32_t>& out, uint16_t& count) {
	uint32_t src = 0;
	count = 0;
	for (uint32_t bit = 0; bit < 32;) {
		if (!((mask >> bit) & 1u)) { ++bit; continue; }
		uint32_t len = 0;
		while (bit + len < 32 && ((mask >> (bit + len)) & 1u)) ++len;
		out.push_back(src | (bit << 8) | (len << 16));
		++count;
		src += len;
		bit += len;
	}
}

bool DeviceTable::set_path(const std::string& path) {
	if (path != "auto" && path != "column" && path != "column_keys" && path != "resident") return false;
	impl_->path = path;
	return true;
}

void DeviceTable::set_l_pref(int l) { impl_->l_pref = std::max(4, std::min(l, RES_LMAX)); }
void DeviceTable::set_lanes(int n) { impl_->max_lanes = n < 1 ? 1 : (n > 16 ? 16 : n); }

void DeviceTable::set_fold(bool v) { impl_->fold = v; }

whamd_status_t DeviceTable::upload(const Problem& p, int device, std::string& msg) {
	Impl& m = *impl_;
	m.device = device;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		msg = "no HIP device visible: the whatshap_amd device path needs an MI355X (gfx950); there is no CPU fallback";
		return WHAMD_ERR_DEVICE;
	}
	if (device < 0 || device >= ndev) {
		msg = "device index " + std::to_string(device) + " out of range (" + std::to_string(ndev) + " visible)";
		return WHAMD_ERR_DEVICE;
	}
	HIP_TRY(hipSetDevice(device));
	if (!m.stream) HIP_TRY(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
	if (!m.ev0) {
		HIP_TRY(hipEventCreate(&m.ev0));
		HIP_TRY(hipEventCreate(&m.ev1));
		HIP_TRY(hipEventCreate(&m.ev2));
		HIP_TRY(hipEventCreate(&m.ev3));
	}
	m.release();
	const uint32_t n = p.n_cols;
	if (n == 0) return WHAMD_OK;
	if (!select_kernels(p.T, p.n_ind, m.fused, m.keysfn)) {
		msg = "no device kernel for T=" + std::to_string(p.T) + ", individuals=" + std::to_string(p.n_ind);
		return WHAMD_ERR_UNSUPPORTED;
	}
	const bool force_keys = m.path == "column_keys";
	const bool want_resident = m.path == "auto" || m.path == "resident";
	const auto tu0 = std::chrono::steady_clock::now();
	plan_forward(p, want_resident, m.l_pref, m.fold, m.plan);
	const auto tu1 = std::chrono::steady_clock::now();
	if (getenv("WHAMD_DEBUG_PLAN")) {
		for (const Step& st : m.plan.steps) {
			if (st.kind == 0) { fprintf(stderr, "[plan] column %u k=%u b=%u f=%u\n", st.index, p.k[st.index], p.b[st.index], p.f[st.index]); continue; }
			const ResSegment& sgm = m.plan.segments[st.index];
			fprintf(stderr, "[plan] run c0=%u ncols=%u g=%u threads=%u max_l=%u stage_words=%u\n", sgm.c0, sgm.ncols, sgm.g, sgm.threads, sgm.max_l, sgm.stage_words);
			for (uint32_t i = 0; i < sgm.ncols; ++i) {
				const ResColumn& rc = m.plan.columns[sgm.col_off + i];
				const ResBacktrace& rb = m.plan.backtrace[sgm.col_off + i];
				fprintf(stderr, "[plan]   col %u mode=%u nfold=%u Lb=%u Lf=%u ebits=%u epos0=%u nthr=%u stage_off=%u nwords=%u | bt layout=%u n_g=%u n_l=%u\n",
				        sgm.c0 + i, rc.mode, rc.nfold, rc.Lb, rc.Lf, rc.ebits, rc.epos[0], rc.nthr, rc.stage_off, rc.nwords, rb.layout, rb.n_g, rb.n_l);
			}
		}
	}
	const uint32_t tbits = 2 * p.n_triples;
	const uint32_t ni = std::max<uint32_t>(p.n_ind, 1);
	// ---- descriptors
	m.cols.assign(n, DevColumn{});
	std::vector<uint32_t> segs;
	std::vector<uint32_t> term_ptr32((size_t)n * (p.T + 1));
	std::vector<DevTerm> terms(p.terms.size());
	for (size_t i = 0; i < p.terms.size(); ++i) terms[i] = DevTerm{p.terms[i].c, p.terms[i].plus, p.terms[i].minus};
	for (uint32_t c = 0; c < n; ++c) {
		if (p.term_end(c, p.T - 1) - p.term_begin(c, 0) > (uint64_t)COL_MAXTERMS) {
			msg = "more than " + std::to_string(COL_MAXTERMS) + " allele-assignment terms in one column: pedigree too complex for the device path";
			return WHAMD_ERR_UNSUPPORTED;
		}
	}
	if (p.terms.size() >= 0xFFFFFFFFull || (uint64_t)p.col_ptr[n] * ni >= 0xFFFFFFFFull) {
		msg = "problem too large for 32-bit device offsets";
		return WHAMD_ERR_UNSUPPORTED;
	}
	uint64_t bt = 0, seg_bt = 0;
	size_t seg_cursor = 0;
	uint32_t max_f = 0, max_keys_f = 0;
	for (uint32_t c = 0; c < n; ++c) {
		DevColumn& d = m.cols[c];
		d.k = p.k[c];
		d.b = p.b[c];
		d.f = p.f[c];
		d.recomb = p.recomb[c];
		d.delta_off = (uint32_t)(p.col_ptr[c] * ni);
		d.term_off = (uint32_t)((size_t)c * (p.T + 1));
		for (uint32_t t = 0; t <= p.T; ++t) term_ptr32[(size_t)c * (p.T + 1) + t] = (uint32_t)p.term_ptr[(size_t)c * p.T + t];
		d.seg_off = (uint32_t)segs.size();
		const uint32_t kmask = d.k >= 32 ? 0xFFFFFFFFu : ((1u << d.k) - 1u);
		append_segments(p.fwd_mask[c], segs, d.nseg_fwd);
		append_segments(kmask & ~p.fwd_mask[c], segs, d.nseg_end);
		d.ebits = d.k - d.f;
		d.is_last = (c + 1 == n);
		d.eloop = std::min<uint32_t>(d.ebits, QMAX);
		d.nplanes = d.ebits + tbits;
		d.bt_off = bt;
		if (m.plan.col_to_res[c] >= 0) {
			d.mode = 2;
			d.res_idx = (uint32_t)m.plan.col_to_res[c];
			if (seg_cursor < m.plan.segments.size() && m.plan.segments[seg_cursor].c0 == c) {  // first column of a run
				ResSegment& sgm = m.plan.segments[seg_cursor];
				sgm.bt_lo = (uint32_t)bt;
				sgm.bt_hi = (uint32_t)(bt >> 32);
				seg_bt = bt;
				bt += (uint64_t)sgm.stage_words * (1ull << sgm.g) * 8ull;
				++seg_cursor;
			}
			d.bt_off = seg_bt;
		} else {
			const bool fused_ok = !force_keys && !d.is_last && d.f >= 6 && d.ebits <= (uint32_t)QMAX;
			d.mode = fused_ok ? 0u : 1u;
			if (d.mode == 0) bt += (uint64_t)d.nplanes * p.T * (1ull << (d.f - 6)) * 8ull;
			else { bt += (uint64_t)p.T * (1ull << d.f) * 4ull; max_keys_f = std::max(max_keys_f, d.f); }
		}
		bt = (bt + 15ull) & ~15ull;
		max_f = std::max(max_f, d.f);
	}
	m.bt_bytes = bt;
	size_t free_b = 0, total_b = 0;
	HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	const uint64_t need = bt + 2ull * (1ull << max_f) * p.T * 4ull + (1ull << max_keys_f) * p.T * 8ull;
	if (need + (1ull << 30) > free_b) {
		msg = "backtrace arena of " + std::to_string(need >> 20) + " MiB does not fit in free HBM (" + std::to_string(free_b >> 20) + " MiB)";
		return WHAMD_ERR_UNSUPPORTED;
	}
	// ---- allocate + upload
	auto alloc = [&](void** dptr, size_t bytes) -> hipError_t {
		hipError_t e = hipMalloc(dptr, std::max<size_t>(bytes, 16));
		if (e == hipSuccess) m.allocations.push_back(*dptr);
		return e;
	};
	auto up = [&](void** dptr, const void* src, size_t bytes) -> hipError_t {
		hipError_t e = alloc(dptr, bytes);
		if (e != hipSuccess) return e;
		if (bytes) e = hipMemcpyAsync(*dptr, src, bytes, hipMemcpyHostToDevice, m.stream);
		return e;
	};
	std::vector<int32_t> delta_fallback;
	const int32_t* delta_src = p.delta.data();
	size_t delta_count = (size_t)p.col_ptr[n] * p.n_ind;
	if (p.n_ind == 0) { delta_fallback.assign(std::max<size_t>(p.col_ptr[n], 1), 0); delta_src = delta_fallback.data(); delta_count = delta_fallback.size(); }
	void *d_delta, *d_term_ptr, *d_terms, *d_segs, *d_bt, *d_keys, *d_last_keys, *d_rcol, *d_rbt;
	HIP_TRY(up((void**)&m.d_cols, m.cols.data(), m.cols.size() * sizeof(DevColumn)));
	HIP_TRY(up(&d_delta, delta_src, delta_count * sizeof(int32_t)));
	HIP_TRY(up(&d_term_ptr, term_ptr32.data(), term_ptr32.size() * sizeof(uint32_t)));
	HIP_TRY(up(&d_terms, terms.data(), terms.size() * sizeof(DevTerm)));
	HIP_TRY(up(&d_segs, segs.data(), segs.size() * sizeof(uint32_t)));
	HIP_TRY(up(&d_rcol, m.plan.columns.data(), m.plan.columns.size() * sizeof(ResColumn)));
	HIP_TRY(up(&d_rbt, m.plan.backtrace.data(), m.plan.backtrace.size() * sizeof(ResBacktrace)));
	void *d_pcol = nullptr, *d_pterm = nullptr;
	m.plan.ped_columns.resize(m.plan.ped_columns.empty() ? 0 : m.plan.columns.size());
	HIP_TRY(up(&d_pcol, m.plan.ped_columns.data(), m.plan.ped_columns.size() * sizeof(PedColumn)));
	HIP_TRY(up(&d_pterm, m.plan.ped_terms.data(), m.plan.ped_terms.size() * sizeof(PedTerm)));
	m.dp.ped_cols = (const PedColumn*)d_pcol;
	m.dp.ped_terms = (const PedTerm*)d_pterm;
	// ---- jobs (see Impl::Job): connected components made of runs only get their own job
	m.release_lanes();
	{
		Impl::Job final_job;
		final_job.final = true;
		std::vector<Impl::Job> component_jobs;
		const std::vector<uint32_t>& first = m.plan.component_first_step;
		const bool split = m.max_lanes > 1 && first.size() > 1 && !getenv("WHAMD_DEBUG_TIMING");
		if (!split) {
			for (uint32_t si = 0; si < m.plan.steps.size(); ++si) final_job.steps.push_back(si);
		} else {
			for (size_t k = 0; k < first.size(); ++k) {
				const uint32_t s0 = first[k], s1 = k + 1 < first.size() ? first[k + 1] : (uint32_t)m.plan.steps.size();
				Impl::Job* job = &final_job;
				if (k + 1 < first.size()) { component_jobs.emplace_back(); job = &component_jobs.back(); }
				for (uint32_t si = s0; si < s1; ++si) job->steps.push_back(si);
			}
		}
		m.jobs.push_back(std::move(final_job));
		for (Impl::Job& j : component_jobs) m.jobs.push_back(std::move(j));
	}
	{
		m.units.clear();
		for (Impl::Job& job : m.jobs) {
		job.unit_off = (uint32_t)m.units.size();
		for (size_t sj = job.steps.size(); sj-- > 0;) {
			const Step& st = m.plan.steps[job.steps[sj]];
			BtUnit u{};
			u.kind = st.kind;
			if (st.kind == 0) {
				// column step: everything the backtrace needs, so that its chain holds no descriptor load
				const DevColumn& d = m.cols[st.index];
				u.c0 = st.index;
				u.ncols = 1;
				u.g = d.f; u.Lf_last = d.mode; u.stage_words = d.nplanes; u.n_wext = d.ebits;
				u.bt_lo = (uint32_t)d.bt_off; u.bt_hi = (uint32_t)(d.bt_off >> 32);
				u.n_lext = d.nseg_fwd; u.pad0 = d.nseg_end;
				const uint32_t nseg = (uint32_t)d.nseg_fwd + d.nseg_end;
				if (nseg <= (uint32_t)(RES_IOSEG + RES_BT_LRUNS)) {
					uint32_t* dst = u.wext;  // wext[6] and lext[10] are contiguous: 16 run slots
					for (uint32_t i = 0; i < nseg; ++i) dst[i] = segs[d.seg_off + i];
					u.pad1[0] = 1;  // runs are inline
				}
			} else {
				const ResSegment& sgm = m.plan.segments[st.index];
				u.c0 = sgm.c0; u.ncols = sgm.ncols; u.col_off = sgm.col_off; u.g = sgm.g; u.Lf_last = sgm.Lf_last;
				u.stage_words = sgm.stage_words; u.n_wext = sgm.n_wext; u.bt_lo = sgm.bt_lo; u.bt_hi = sgm.bt_hi;
				u.n_lext = sgm.n_lext;
				u.pad0 = (uint32_t)sgm.bt_active | ((uint32_t)sgm.bt_simple << 16);
				std::copy(sgm.wext, sgm.wext + RES_IOSEG, u.wext);
				std::copy(sgm.lext, sgm.lext + RES_BT_LRUNS, u.lext);
			}
			m.units.push_back(u);
		}
		job.unit_count = (uint32_t)m.units.size() - job.unit_off;
		}
	}
	HIP_TRY(up((void**)&m.d_units, m.units.data(), m.units.size() * sizeof(BtUnit)));
	{
		uint32_t max_stage = 0;
		for (const ResSegment& sgm : m.plan.segments) max_stage = std::max(max_stage, sgm.stage_words);
		m.bt_lds = (size_t)2 * RES_MAXCOLS * 128 + 512 + 16 + (size_t)RES_MAXCOLS * 8 + (size_t)max_stage * 8 + 16;
	}
	void* d_rtab = nullptr;
	const bool ped_plan = !m.plan.ped_columns.empty();
	HIP_TRY(alloc(&d_rtab, m.plan.columns.size() * (ped_plan ? PED_TABLE : RES_TABLE) * sizeof(int32_t)));
	m.dp.res_tables = ped_plan ? nullptr : (int32_t*)d_rtab;
	m.dp.ped_tables = ped_plan ? (int32_t*)d_rtab : nullptr;
	const auto tu2 = std::chrono::steady_clock::now();
	HIP_TRY(alloc(&d_bt, bt));
	const auto tu3 = std::chrono::steady_clock::now();
	m.key_entries = (size_t)(1ull << max_keys_f) * p.T;
	HIP_TRY(alloc(&d_keys, m.key_entries * 8));
	HIP_TRY(alloc(&d_last_keys, (size_t)MAX_T * 8));
	HIP_TRY(alloc((void**)&m.d_pr[0], (size_t)(1ull << max_f) * p.T * 4));
	HIP_TRY(alloc((void**)&m.d_pr[1], (size_t)(1ull << max_f) * p.T * 4));
	HIP_TRY(alloc((void**)&m.d_path_index, (size_t)n * 4));
	HIP_TRY(alloc((void**)&m.d_path_trans, (size_t)n * 4));
	HIP_TRY(alloc((void**)&m.d_score, 16));
	HIP_TRY(hipHostMalloc((void**)&m.h_pinned, (2 * (size_t)n + 4 + m.jobs.size()) * sizeof(uint32_t), hipHostMallocDefault));
	HIP_TRY(alloc((void**)&m.d_job_scores, (m.jobs.size() + 1) * 4));
	{   // lanes: longest job first to the least loaded lane; lane 0 always runs the final job
		const size_t n_lanes = std::max<size_t>(1, std::min<size_t>((size_t)m.max_lanes, m.jobs.size()));
		m.lanes.assign(n_lanes, Impl::Lane());
		std::vector<uint64_t> load(n_lanes, 0);
		std::vector<uint32_t> order;
		for (uint32_t j = 1; j < m.jobs.size(); ++j) order.push_back(j);
		std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return m.jobs[a].steps.size() > m.jobs[b].steps.size(); });
		m.lanes[0].jobs.push_back(0);
		load[0] = m.jobs[0].steps.size() + 1;
		for (uint32_t j : order) {
			const size_t l = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
			m.lanes[l].jobs.push_back(j);
			load[l] += m.jobs[j].steps.size() + 1;
		}
		m.lanes[0].stream = m.stream;
		m.lanes[0].d_pr[0] = m.d_pr[0];
		m.lanes[0].d_pr[1] = m.d_pr[1];
		for (size_t l = 1; l < n_lanes; ++l) {
			HIP_TRY(hipStreamCreateWithFlags(&m.lanes[l].stream, hipStreamNonBlocking));
			HIP_TRY(hipEventCreateWithFlags(&m.lanes[l].done, hipEventDisableTiming));
			HIP_TRY(alloc((void**)&m.lanes[l].d_pr[0], (size_t)(1ull << max_f) * p.T * 4));
			HIP_TRY(alloc((void**)&m.lanes[l].d_pr[1], (size_t)(1ull << max_f) * p.T * 4));
			HIP_TRY(alloc((void**)&m.lanes[l].d_keys, m.key_entries * 8));
		}
		m.lanes[0].d_keys = (unsigned long long*)d_keys;
		if (!m.ev_ready) HIP_TRY(hipEventCreateWithFlags(&m.ev_ready, hipEventDisableTiming));
	}
	HIP_TRY(hipStreamSynchronize(m.stream));
	m.dp.cols = m.d_cols;
	m.dp.delta = (const int32_t*)d_delta;
	m.dp.term_ptr = (const uint32_t*)d_term_ptr;
	m.dp.terms = (const DevTerm*)d_terms;
	m.dp.segs = (const uint32_t*)d_segs;
	m.dp.bt = (uint8_t*)d_bt;
	m.dp.keys = (unsigned long long*)d_keys;
	m.dp.last_keys = (unsigned long long*)d_last_keys;
	m.dp.res_cols = (const ResColumn*)d_rcol;
	m.dp.res_bt = (const ResBacktrace*)d_rbt;
	m.dp.dbg = nullptr;
	if (getenv("WHAMD_DEBUG_TIMING")) {
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		fprintf(stderr, "[whamd timing] upload: plan %.1f ms, descriptors + copies %.1f ms, backtrace arena (%.2f GB) %.1f ms, rest %.1f ms\n",
		        ms(tu0, tu1), ms(tu1, tu2), (double)bt / 1e9, ms(tu2, tu3), ms(tu3, std::chrono::steady_clock::now()));
	}
	if (getenv("WHAMD_DEBUG_TIMING")) {
		void* d_dbg = nullptr;
		const size_t dbg_bytes = (m.plan.segments.size() + 1) * 64 + 4 * 512 * 16 + 64;
		HIP_TRY(alloc(&d_dbg, dbg_bytes));
		HIP_TRY(hipMemset(d_dbg, 0, dbg_bytes));
		m.dp.dbg = (unsigned long long*)d_dbg;
		m.dp.dbg_wg_off = (uint32_t)((m.plan.segments.size() + 1) * 8);
		m.dp.dbg_flags = (uint32_t)atoi(getenv("WHAMD_DEBUG_TIMING"));
	}
	m.dp.n_cols = n;
	m.dp.T = p.T;
	m.dp.tbits = tbits;
	m.dp.n_ind = p.n_ind;
	// kernels with more than 64 KiB of dynamic LDS need the opt-in on every device they run on
	HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_segment<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_segment<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(backtrace_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_segment_ped<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_segment_ped<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	return WHAMD_OK;
}

whamd_status_t DeviceTable::solve(const Problem& p, Solution& s, whamd_solve_stats& st, std::string& msg) {
	whamd_status_t status = enqueue(p, s, msg);
	if (status != WHAMD_OK) return status;
	return wait(p, s, st, msg);
}

whamd_status_t DeviceTable::enqueue(const Problem& p, Solution& s, std::string& msg) {
	bool done = false;
	whamd_status_t status = WHAMD_OK;
	while (status == WHAMD_OK && !done) status = enqueue_some(p, s, ~0ull, done, msg);
	return status;
}

// Launches one forward step on a lane's stream: reads `prev`, writes `cur`.
void DeviceTable::Impl::launch_step(const Problem& p, const Step& step, const Lane& lane, const uint32_t* prev, uint32_t* cur,
                                    uint64_t& launches) {
	Impl& m = *this;
	hipStream_t stream = lane.stream;
	DevProblem dp = m.dp;
	dp.keys = lane.d_keys;
	if (step.kind == 1) {
		const ResSegment& sg = m.plan.segments[step.index];
		if (sg.kind == 1) {
			const size_t words = ((size_t)sg.ncols * (PED_LDSWORDS + PED_TABLE) + (size_t)sg.n_terms * 2 + 3) & ~(size_t)3;
			const size_t lds_ped = words * 4 + 2 * ((size_t)16 << sg.max_l) + (size_t)sg.stage_words * 8;
			ResSegment parg = sg;
			parg.pad = step.index;
			if (m.dp.dbg) hipLaunchKernelGGL(resident_segment_ped<true>, dim3(1u << sg.g), dim3(sg.threads), lds_ped, stream, dp, parg, prev, cur);
			else hipLaunchKernelGGL(resident_segment_ped<false>, dim3(1u << sg.g), dim3(sg.threads), lds_ped, stream, dp, parg, prev, cur);
			launches += 1;
			return;
		}
		const size_t lds = (size_t)sg.ncols * (64 + RES_TABLE) * 4 + 2 * ((size_t)4 << sg.max_l) + (size_t)sg.stage_words * 8;
		ResSegment arg = sg;
		arg.pad = step.index;
		if (m.dp.dbg) hipLaunchKernelGGL(resident_segment<true>, dim3(1u << sg.g), dim3(sg.threads), lds, stream, dp, arg, prev, cur);
		else hipLaunchKernelGGL(resident_segment<false>, dim3(1u << sg.g), dim3(sg.threads), lds, stream, dp, arg, prev, cur);
		launches += 1;
		return;
	}
	const uint32_t c = step.index;
	const DevColumn& d = m.cols[c];
	if (d.mode == 0) {
		const uint32_t threads = 1u << d.f;
		const uint32_t block = std::min<uint32_t>(256, threads);
		hipLaunchKernelGGL(m.fused, dim3(threads / block), dim3(block), 0, stream, dp, c, prev, cur);
		launches += 1;
	} else {
		const uint64_t total = 1ull << (d.f + d.ebits - d.eloop);
		const uint32_t block = (uint32_t)std::min<uint64_t>(256, (total + 63) / 64 * 64);
		hipLaunchKernelGGL(m.keysfn, dim3((uint32_t)((total + block - 1) / block)), dim3(block), 0, stream, dp, c, prev, (uint32_t)total);
		const uint32_t entries = (1u << d.f) * p.T;
		const uint32_t fblock = std::min<uint32_t>(256, (entries + 63) / 64 * 64);
		hipLaunchKernelGGL(column_finalize, dim3((entries + fblock - 1) / fblock), dim3(fblock), 0, stream, dp, c, cur, entries);
		launches += 2;
	}
}

// Submits up to `max_launches` launches of lane `li` (forward steps of its jobs in order; after a job's last step its
// score copy and backtrace).  `finished`: the lane has nothing left.
whamd_status_t DeviceTable::Impl::submit_lane(const Problem& p, size_t li, uint64_t max_launches, uint64_t& launches, bool& finished,
                                              std::string& msg) {
	Impl& m = *this;
	Impl::Lane& lane = m.lanes[li];
	uint64_t in_turn = 0;
	while (lane.job_i < lane.jobs.size() && in_turn < max_launches) {
		const Impl::Job& job = m.jobs[lane.jobs[lane.job_i]];
		const uint32_t job_id = lane.jobs[lane.job_i];
		if (lane.step_i == 0)  // a job starts from cost 0: the single entry the first step may read
			HIP_TRY(hipMemsetAsync(lane.d_pr[lane.flip], 0, 4 * (size_t)p.T, lane.stream));
		if (lane.step_i < job.steps.size()) {
			const uint32_t si = job.steps[lane.step_i++];
			uint64_t issued = 0;
			m.launch_step(p, m.plan.steps[si], lane, lane.d_pr[lane.flip], lane.d_pr[lane.flip ^ 1], issued);
			lane.flip ^= 1;
			launches += issued;
			in_turn += issued;
			if (lane.step_i < job.steps.size()) continue;
		}
		// ---- the job's forward pass is submitted: its score (components) and its backtrace
		HIP_TRY(hipGetLastError());
		if (job.final) {
			HIP_TRY(hipEventRecord(m.ev1, lane.stream));
		} else {
			HIP_TRY(hipMemcpyAsync(m.d_job_scores + job_id, lane.d_pr[lane.flip], 4, hipMemcpyDeviceToDevice, lane.stream));
		}
		if (job.unit_count)
			hipLaunchKernelGGL(backtrace_kernel, dim3(1), dim3(1024), m.bt_lds, lane.stream, m.dp, m.d_units + job.unit_off, job.unit_count,
			                   job.final ? 1u : 0u, m.d_path_index, m.d_path_trans, m.d_score);
		HIP_TRY(hipGetLastError());
		if (job.final) HIP_TRY(hipEventRecord(m.ev2, lane.stream));
		++lane.job_i;
		lane.step_i = 0;
		in_turn += 1;
	}
	if (lane.job_i == lane.jobs.size() && lane.step_i != ~(size_t)0) {
		lane.step_i = ~(size_t)0;  // lane finished (marker)
		if (li) HIP_TRY(hipEventRecord(lane.done, lane.stream));
		if (m.lanes_open) --m.lanes_open;
	}
	finished = lane.job_i == lane.jobs.size();
	return WHAMD_OK;
}

// Resumable submission: the first call does the preamble, every call submits at most `budget` forward launches -- round
// robin over the lanes, a few launches per lane and turn, so that the lanes' streams fill up side by side -- and the
// call that runs out of work joins the lanes and appends the downloads.  Lets one host thread interleave the launch
// sequences of several tables as well (whamd_dptable_enqueue_many).
whamd_status_t DeviceTable::enqueue_some(const Problem& p, Solution& s, uint64_t budget, bool& done, std::string& msg) {
	Impl& m = *impl_;
	const uint32_t n = p.n_cols;
	done = false;
	if (n) HIP_TRY(hipSetDevice(m.device));
	if (!m.enqueue_open) {
		m.enqueue_open = true;
		s.path_index.assign(n, 0);
		s.path_trans.assign(n, 0);
		m.launches = 0;
		if (n == 0) {  // src/pedigreedptable.cpp:88-92
			s.optimal_score = 0;
			m.enqueue_open = false;
			done = true;
			return WHAMD_OK;
		}
		for (const Impl::Lane& lane : m.lanes) HIP_TRY(hipMemsetAsync(lane.d_keys, 0xFF, m.key_entries * 8, m.stream));
		HIP_TRY(hipMemsetAsync(m.dp.last_keys, 0xFF, (size_t)MAX_T * 8, m.stream));
		HIP_TRY(hipEventRecord(m.ev0, m.stream));
		if (!m.plan.ped_columns.empty()) {
			const uint32_t entries = (uint32_t)m.plan.ped_columns.size() * PED_TABLE;
			hipLaunchKernelGGL(ped_tables, dim3((entries + 255) / 256), dim3(256), 0, m.stream, m.dp.ped_cols, (uint32_t)m.plan.ped_columns.size(), m.dp.ped_tables);
		} else if (!m.plan.columns.empty()) {
			const uint32_t entries = (uint32_t)m.plan.columns.size() * RES_TABLE;
			hipLaunchKernelGGL(resident_tables, dim3((entries + 255) / 256), dim3(256), 0, m.stream, m.dp.res_cols, (uint32_t)m.plan.columns.size(), m.dp.res_tables);
		}
		if (m.lanes.size() > 1) {
			HIP_TRY(hipEventRecord(m.ev_ready, m.stream));
			for (size_t l = 1; l < m.lanes.size(); ++l) HIP_TRY(hipStreamWaitEvent(m.lanes[l].stream, m.ev_ready, 0));
		}
		for (Impl::Lane& lane : m.lanes) { lane.job_i = 0; lane.step_i = 0; lane.flip = 0; }
		m.lanes_open = m.lanes.size();
	}
	constexpr uint64_t SLICE = 16;
	uint64_t launches = 0;
	while (m.lanes_open && launches < budget) {
		for (size_t li = 0; li < m.lanes.size() && launches < budget; ++li) {
			bool finished = false;
			uint64_t issued = 0;
			const whamd_status_t st = m.submit_lane(p, li, std::min<uint64_t>(SLICE, budget - launches), issued, finished, msg);
			if (st != WHAMD_OK) return st;
			launches += issued;
		}
	}
	m.launches += launches;
	if (m.lanes_open) return WHAMD_OK;
	for (size_t l = 1; l < m.lanes.size(); ++l) HIP_TRY(hipStreamWaitEvent(m.stream, m.lanes[l].done, 0));
	// downloads go to pinned host buffers: a copy into pageable memory would block this call until the stream drains
	HIP_TRY(hipMemcpyAsync(m.h_pinned, m.d_path_index, (size_t)n * 4, hipMemcpyDeviceToHost, m.stream));
	HIP_TRY(hipMemcpyAsync(m.h_pinned + n, m.d_path_trans, (size_t)n * 4, hipMemcpyDeviceToHost, m.stream));
	HIP_TRY(hipMemcpyAsync(m.h_pinned + 2 * (size_t)n, m.d_score, 4, hipMemcpyDeviceToHost, m.stream));
	if (m.jobs.size() > 1)
		HIP_TRY(hipMemcpyAsync(m.h_pinned + 2 * (size_t)n + 1, m.d_job_scores + 1, (m.jobs.size() - 1) * 4, hipMemcpyDeviceToHost, m.stream));
	HIP_TRY(hipEventRecord(m.ev3, m.stream));
	m.enqueue_open = false;
	done = true;
	return WHAMD_OK;
}

whamd_status_t DeviceTable::wait(const Problem& p, Solution& s, whamd_solve_stats& st, std::string& msg) {
	Impl& m = *impl_;
	if (p.n_cols == 0) return WHAMD_OK;
	HIP_TRY(hipSetDevice(m.device));
	const uint64_t launches = m.launches;
	HIP_TRY(hipStreamSynchronize(m.stream));
	const uint32_t n = p.n_cols;
	std::memcpy(s.path_index.data(), m.h_pinned, (size_t)n * 4);
	std::memcpy(s.path_trans.data(), m.h_pinned + n, (size_t)n * 4);
	s.optimal_score = m.h_pinned[2 * (size_t)n];
	for (size_t j = 1; j < m.jobs.size(); ++j) s.optimal_score += m.h_pinned[2 * (size_t)n + j];  // connected components solved as their own jobs
	float f01 = 0, f12 = 0, f03 = 0;
	HIP_TRY(hipEventElapsedTime(&f01, m.ev0, m.ev1));
	HIP_TRY(hipEventElapsedTime(&f12, m.ev1, m.ev2));
	HIP_TRY(hipEventElapsedTime(&f03, m.ev0, m.ev3));
	st.forward_ms = f01;
	st.backtrace_ms = f12;
	st.total_ms = f03;
	st.forward_launches = launches;
	if (m.dp.dbg) {
		std::vector<unsigned long long> d(m.plan.segments.size() * 8);
		HIP_TRY(hipMemcpy(d.data(), m.dp.dbg, d.size() * 8, hipMemcpyDeviceToHost));
		unsigned long long a = 0, b = 0, c2 = 0, cols = 0, p1 = 0, p2 = 0, p3 = 0, ns = 0;
		for (size_t i = 0; i < m.plan.segments.size(); ++i) { a += d[8 * i]; b += d[8 * i + 1]; c2 += d[8 * i + 2]; cols += d[8 * i + 3]; p1 += d[8 * i + 4]; p2 += d[8 * i + 5]; p3 += d[8 * i + 6]; ns += d[8 * i + 7]; }
		if (!m.plan.ped_columns.empty() || (m.dp.dbg_flags & 4u))
			fprintf(stderr, "[whamd timing] run prologue (wave 0 of workgroup 0), cycles after the first instruction: kernel arguments usable %.0f, first loaded data %.0f, everything staged %.0f\n",
			        (double)p2 / std::max<unsigned long long>(ns, 1), (double)p3 / std::max<unsigned long long>(ns, 1), (double)p1 / std::max<unsigned long long>(ns, 1));
		else
		fprintf(stderr, "[whamd timing] per barrier step (wave 0 of workgroup 0, %.1f steps per run): hot words %.0f, evaluate %.0f, barrier %.0f cycles\n",
		        (double)ns / m.plan.segments.size(), (double)p1 / std::max<unsigned long long>(ns, 1), (double)p2 / std::max<unsigned long long>(ns, 1), (double)p3 / std::max<unsigned long long>(ns, 1));
		{
			unsigned long long b3[6] = {0, 0, 0, 0, 0, 0};
			HIP_TRY(hipMemcpy(b3, m.dp.dbg + m.dp.dbg_wg_off + 4 * 512 * 2, sizeof b3, hipMemcpyDeviceToHost));
			if (b3[2]) fprintf(stderr, "[whamd timing] backtrace per run: record load + prefetch %.0f cycles, walk + hand-over %.0f cycles (%llu runs); of the latter: local exit index %.0f, chain %.0f, logical indices + stores %.0f\n",
			                   (double)b3[0] / b3[2], (double)b3[1] / b3[2], b3[2], (double)b3[3] / b3[2], (double)b3[4] / b3[2], (double)b3[5] / b3[2]);
		}
		if (m.plan.segments.size() > 104) {
			std::vector<unsigned long long> wg(4 * 512 * 2);
			HIP_TRY(hipMemcpy(wg.data(), m.dp.dbg + m.dp.dbg_wg_off, wg.size() * 8, hipMemcpyDeviceToHost));
			unsigned long long prev_end = 0;
			for (int sgi = 0; sgi < 4; ++sgi) {
				const uint32_t G = 1u << m.plan.segments[100 + sgi].g;
				unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
				for (uint32_t ww = 0; ww < G; ++ww) {
					const unsigned long long a2 = wg[((size_t)sgi * 512 + ww) * 2], b2 = wg[((size_t)sgi * 512 + ww) * 2 + 1];
					s0 = std::min(s0, a2); s1 = std::max(s1, a2); e0 = std::min(e0, b2); e1 = std::max(e1, b2);
				}
				fprintf(stderr, "[whamd timing] run %d (%u workgroups): first start +%.2f us after previous run's last end; starts spread %.2f us; first end %.2f us, last end %.2f us after first start\n",
				        100 + sgi, G, prev_end ? (double)(s0 - prev_end) / 100.0 : 0.0, (double)(s1 - s0) / 100.0, (double)(e0 - s0) / 100.0, (double)(e1 - s0) / 100.0);
				prev_end = e1;
			}
		}
		fprintf(stderr, "[whamd timing] segments %zu cols %llu | cycles/segment: prologue %.0f columns %.0f (%.0f per column) store %.0f | fwd %.3f ms, %.2f us per segment\n",
		        m.plan.segments.size(), cols, (double)a / m.plan.segments.size(), (double)b / m.plan.segments.size(),
		        (double)b / std::max<unsigned long long>(cols, 1), (double)c2 / m.plan.segments.size(), f01, f01 * 1e3 / m.plan.segments.size());
	}
	return WHAMD_OK;
}

}  // namespace whamd
