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

// cost_{c,t}(x) for all t (get_cost, src/pedigreecolumncostcomputer.cpp:101-114) from the per-individual sums L_s(x).
template <int T, int NIND>
__device__ __forceinline__ void cell_costs(uint32_t x, const DevProblem& P, const DevColumn& col, uint32_t (&cost)[T]) {
	int32_t L[NIND];
#pragma unroll
	for (int s = 0; s < NIND; ++s) L[s] = 0;
	const int32_t* __restrict__ dl = P.delta + col.delta_off;
	const uint32_t k = col.k;
	for (uint32_t j = 0; j < k; ++j) {
		const bool bit = (x >> j) & 1u;
#pragma unroll
		for (int s = 0; s < NIND; ++s) L[s] += bit ? dl[s * k + j] : 0;
	}
	const uint32_t* __restrict__ tp = P.term_ptr + col.term_off;
#pragma unroll
	for (int t = 0; t < T; ++t) {
		uint32_t best = 0xFFFFFFFFu;
		const uint32_t e = tp[t + 1];
		for (uint32_t q = tp[t]; q < e; ++q) {
			const DevTerm tm = P.terms[q];
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
		cell_costs<T, NIND>(x, P, col, cost);
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
		cell_costs<T, NIND>(x, P, col, cost);
		cell_dp<T>(cost, pr, x & lowmask, col.recomb, D, aj);
		const uint32_t r = gray_rank(x);
#pragma unroll
		for (int i = 0; i < T; ++i) {
			const unsigned long long key = ((unsigned long long)D[i] << 32) | ((unsigned long long)r << 4) | aj[i];
			best[i] = min(best[i], key);
		}
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

// Backtrace (src/pedigreedptable.cpp:137-173) by one lane; out: index / transmission per column, out_score[0] = optimum.
__global__ void backtrace_kernel(DevProblem P, uint32_t* __restrict__ path_index, uint32_t* __restrict__ path_trans,
                                 uint32_t* __restrict__ out_score) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const uint32_t n = P.n_cols, T = P.T;
	// optimum of the last column: first (rank(x), i) attaining the minimum (strict '<' scan, :306-315)
	unsigned long long bestk = ~0ull;
	uint32_t t = 0, tprev = 0;
	for (uint32_t i = 0; i < T; ++i) {
		const unsigned long long key = P.last_keys[i];
		if ((key >> 4) < (bestk >> 4)) { bestk = key; t = i; }
	}
	if (bestk == ~0ull) {  // unreachable for valid inputs (the host rejects Mendelian conflicts); keep defined output
		out_score[0] = 0xFFFFFFFFu;
		bestk = 0;
	} else {
		out_score[0] = (uint32_t)(bestk >> 32);
	}
	const uint32_t rlast = (uint32_t)(bestk >> 4) & 0x0FFFFFFFu;
	uint32_t x = rlast ^ (rlast >> 1);
	tprev = (uint32_t)bestk & 15u;
	path_index[n - 1] = x;
	path_trans[n - 1] = t;
	for (uint32_t c = n - 1; c > 0; --c) {
		const DevColumn cc = P.cols[c];
		const DevColumn pc = P.cols[c - 1];
		const uint32_t y = x & ((1u << cc.b) - 1u);
		uint32_t xp, aj;
		if (pc.mode == 0) {
			const unsigned long long* planes = reinterpret_cast<const unsigned long long*>(P.bt + pc.bt_off);
			const uint32_t words = 1u << (pc.f - 6);
			uint32_t v = 0;
			for (uint32_t p = 0; p < pc.nplanes; ++p) {
				const unsigned long long word = planes[(size_t)(p * T + tprev) * words + (y >> 6)];
				v |= (uint32_t)((word >> (y & 63u)) & 1ull) << p;
			}
			const uint32_t e = v & ((1u << pc.ebits) - 1u);
			aj = v >> pc.ebits;
			const uint32_t* segs = P.segs + pc.seg_off;
			xp = deposit(y, segs, pc.nseg_fwd) | deposit(e, segs + pc.nseg_fwd, pc.nseg_end);
		} else {
			const uint32_t raw = reinterpret_cast<const uint32_t*>(P.bt + pc.bt_off)[(size_t)y * T + tprev];
			const uint32_t r = raw >> 4;
			xp = r ^ (r >> 1);
			aj = raw & 15u;
		}
		path_index[c - 1] = xp;
		path_trans[c - 1] = tprev;
		tprev = aj;
		x = xp;
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
	// device allocations
	DevColumn* d_cols = nullptr;
	int32_t* d_delta = nullptr;
	uint32_t* d_term_ptr = nullptr;
	DevTerm* d_terms = nullptr;
	uint32_t* d_segs = nullptr;
	uint8_t* d_bt = nullptr;
	unsigned long long* d_keys = nullptr;
	unsigned long long* d_last_keys = nullptr;
	uint32_t* d_pr[2] = {nullptr, nullptr};
	uint32_t* d_path_index = nullptr;
	uint32_t* d_path_trans = nullptr;
	uint32_t* d_score = nullptr;
	std::vector<DevColumn> cols;
	DevProblem dp{};
	FusedFn fused = nullptr;
	KeysFn keysfn = nullptr;
	size_t key_entries = 0;
	bool force_keys = false;
	uint64_t bt_bytes = 0;

	void release() {
		if (d_cols) (void)hipFree(d_cols);
		if (d_delta) (void)hipFree(d_delta);
		if (d_term_ptr) (void)hipFree(d_term_ptr);
		if (d_terms) (void)hipFree(d_terms);
		if (d_segs) (void)hipFree(d_segs);
		if (d_bt) (void)hipFree(d_bt);
		if (d_keys) (void)hipFree(d_keys);
		if (d_last_keys) (void)hipFree(d_last_keys);
		if (d_pr[0]) (void)hipFree(d_pr[0]);
		if (d_pr[1]) (void)hipFree(d_pr[1]);
		if (d_path_index) (void)hipFree(d_path_index);
		if (d_path_trans) (void)hipFree(d_path_trans);
		if (d_score) (void)hipFree(d_score);
		d_cols = nullptr; d_delta = nullptr; d_term_ptr = nullptr; d_terms = nullptr; d_segs = nullptr; d_bt = nullptr;
		d_keys = nullptr; d_last_keys = nullptr; d_pr[0] = d_pr[1] = nullptr; d_path_index = d_path_trans = d_score = nullptr;
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
		if (impl_->stream) (void)hipStreamDestroy(impl_->stream);
		delete impl_;
	}
}

int DeviceTable::device_count() {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// Runs of set bits of `mask` as deposit segments; `src` counts the bits of the compact value consumed so far.
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
	const uint32_t tbits = 2 * p.n_triples;
	const uint32_t ni = std::max<uint32_t>(p.n_ind, 1);
	// ---- descriptors
	m.cols.assign(n, DevColumn{});
	std::vector<uint32_t> segs;
	std::vector<uint32_t> term_ptr32((size_t)n * (p.T + 1));
	std::vector<DevTerm> terms(p.terms.size());
	for (size_t i = 0; i < p.terms.size(); ++i) terms[i] = DevTerm{p.terms[i].c, p.terms[i].plus, p.terms[i].minus};
	if (p.terms.size() >= 0xFFFFFFFFull || (uint64_t)p.col_ptr[n] * ni >= 0xFFFFFFFFull) {
		msg = "problem too large for 32-bit device offsets";
		return WHAMD_ERR_UNSUPPORTED;
	}
	uint64_t bt = 0;
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
		const bool fused_ok = !m.force_keys && !d.is_last && d.f >= 6 && d.ebits <= (uint32_t)QMAX;
		d.mode = fused_ok ? 0u : 1u;
		d.eloop = std::min<uint32_t>(d.ebits, QMAX);
		d.nplanes = d.ebits + tbits;
		d.bt_off = bt;
		if (d.mode == 0) bt += (uint64_t)d.nplanes * p.T * (1ull << (d.f - 6)) * 8ull;
		else { bt += (uint64_t)p.T * (1ull << d.f) * 4ull; max_keys_f = std::max(max_keys_f, d.f); }
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
	auto up = [&](auto*& dptr, const void* src, size_t bytes) -> hipError_t {
		hipError_t e = hipMalloc((void**)&dptr, std::max<size_t>(bytes, 16));
		if (e != hipSuccess) return e;
		if (bytes) e = hipMemcpyAsync(dptr, src, bytes, hipMemcpyHostToDevice, m.stream);
		return e;
	};
	HIP_TRY(up(m.d_cols, m.cols.data(), m.cols.size() * sizeof(DevColumn)));
	std::vector<int32_t> delta_fallback;
	const int32_t* delta_src = p.delta.data();
	size_t delta_count = (size_t)p.col_ptr[n] * p.n_ind;
	if (p.n_ind == 0) { delta_fallback.assign(std::max<size_t>(p.col_ptr[n], 1), 0); delta_src = delta_fallback.data(); delta_count = delta_fallback.size(); }
	HIP_TRY(up(m.d_delta, delta_src, delta_count * sizeof(int32_t)));
	HIP_TRY(up(m.d_term_ptr, term_ptr32.data(), term_ptr32.size() * sizeof(uint32_t)));
	HIP_TRY(up(m.d_terms, terms.data(), terms.size() * sizeof(DevTerm)));
	HIP_TRY(up(m.d_segs, segs.data(), segs.size() * sizeof(uint32_t)));
	HIP_TRY(hipMalloc((void**)&m.d_bt, std::max<uint64_t>(bt, 16)));
	m.key_entries = (size_t)(1ull << max_keys_f) * p.T;
	HIP_TRY(hipMalloc((void**)&m.d_keys, m.key_entries * 8));
	HIP_TRY(hipMalloc((void**)&m.d_last_keys, (size_t)MAX_T * 8));
	HIP_TRY(hipMalloc((void**)&m.d_pr[0], (size_t)(1ull << max_f) * p.T * 4));
	HIP_TRY(hipMalloc((void**)&m.d_pr[1], (size_t)(1ull << max_f) * p.T * 4));
	HIP_TRY(hipMalloc((void**)&m.d_path_index, (size_t)n * 4));
	HIP_TRY(hipMalloc((void**)&m.d_path_trans, (size_t)n * 4));
	HIP_TRY(hipMalloc((void**)&m.d_score, 16));
	HIP_TRY(hipStreamSynchronize(m.stream));
	m.dp.cols = m.d_cols;
	m.dp.delta = m.d_delta;
	m.dp.term_ptr = m.d_term_ptr;
	m.dp.terms = m.d_terms;
	m.dp.segs = m.d_segs;
	m.dp.bt = m.d_bt;
	m.dp.keys = m.d_keys;
	m.dp.last_keys = m.d_last_keys;
	m.dp.n_cols = n;
	m.dp.T = p.T;
	m.dp.tbits = tbits;
	m.dp.n_ind = p.n_ind;
	return WHAMD_OK;
}

void DeviceTable::set_force_keys(bool v) { impl_->force_keys = v; }

whamd_status_t DeviceTable::solve(const Problem& p, Solution& s, whamd_solve_stats& st, std::string& msg) {
	Impl& m = *impl_;
	const uint32_t n = p.n_cols;
	s.path_index.assign(n, 0);
	s.path_trans.assign(n, 0);
	if (n == 0) {  // src/pedigreedptable.cpp:88-92
		s.optimal_score = 0;
		return WHAMD_OK;
	}
	HIP_TRY(hipSetDevice(m.device));
	HIP_TRY(hipMemsetAsync(m.d_keys, 0xFF, m.key_entries * 8, m.stream));
	HIP_TRY(hipMemsetAsync(m.d_last_keys, 0xFF, (size_t)MAX_T * 8, m.stream));
	HIP_TRY(hipEventRecord(m.ev0, m.stream));
	uint64_t launches = 0;
	for (uint32_t c = 0; c < n; ++c) {
		const DevColumn& d = m.cols[c];
		const uint32_t* prev = m.d_pr[(c + 1) & 1];
		uint32_t* cur = m.d_pr[c & 1];
		if (d.mode == 0) {
			const uint32_t threads = 1u << d.f;
			const uint32_t block = std::min<uint32_t>(256, threads);
			hipLaunchKernelGGL(m.fused, dim3(threads / block), dim3(block), 0, m.stream, m.dp, c, prev, cur);
			++launches;
		} else {
			const uint64_t total = 1ull << (d.f + d.ebits - d.eloop);
			const uint32_t block = (uint32_t)std::min<uint64_t>(256, (total + 63) / 64 * 64);
			hipLaunchKernelGGL(m.keysfn, dim3((uint32_t)((total + block - 1) / block)), dim3(block), 0, m.stream, m.dp, c, prev, (uint32_t)total);
			const uint32_t entries = (1u << d.f) * p.T;
			const uint32_t fblock = std::min<uint32_t>(256, (entries + 63) / 64 * 64);
			hipLaunchKernelGGL(column_finalize, dim3((entries + fblock - 1) / fblock), dim3(fblock), 0, m.stream, m.dp, c, cur, entries);
			launches += 2;
		}
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(m.ev1, m.stream));
	hipLaunchKernelGGL(backtrace_kernel, dim3(1), dim3(64), 0, m.stream, m.dp, m.d_path_index, m.d_path_trans, m.d_score);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(m.ev2, m.stream));
	HIP_TRY(hipMemcpyAsync(s.path_index.data(), m.d_path_index, (size_t)n * 4, hipMemcpyDeviceToHost, m.stream));
	HIP_TRY(hipMemcpyAsync(s.path_trans.data(), m.d_path_trans, (size_t)n * 4, hipMemcpyDeviceToHost, m.stream));
	uint32_t score = 0;
	HIP_TRY(hipMemcpyAsync(&score, m.d_score, 4, hipMemcpyDeviceToHost, m.stream));
	HIP_TRY(hipEventRecord(m.ev3, m.stream));
	HIP_TRY(hipStreamSynchronize(m.stream));
	s.optimal_score = score;
	float f01 = 0, f12 = 0, f03 = 0;
	HIP_TRY(hipEventElapsedTime(&f01, m.ev0, m.ev1));
	HIP_TRY(hipEventElapsedTime(&f12, m.ev1, m.ev2));
	HIP_TRY(hipEventElapsedTime(&f03, m.ev0, m.ev3));
	st.forward_ms = f01;
	st.backtrace_ms = f12;
	st.total_ms = f03;
	st.forward_launches = launches;
	return WHAMD_OK;
}

}  // namespace whamd
