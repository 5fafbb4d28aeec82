// heuristic_device.hip -- the PedMecHeuristic beam search on the device: heuristic_core.h instantiated as ONE persistent
// single-workgroup kernel (1024 threads, one launch per table).  The algorithm is a chain over the columns and, inside a column,
// over the reads that start there; what is parallel is the beam (row_limit .. 65535 x 4^trios solutions): every phase of a
// column -- projection + duplicate merging (hash table with atomics), the two placements of a read (independent float work per
// solution), the pruning (radix select of the threshold, ordered compaction), the alternative transmission values, the phasing
// cost -- runs over the solutions with the workgroup's threads and meets at workgroup barriers.  No launch boundary, no host
// round trip inside the table; the solution pools live in HBM (L2-resident at the default row limit).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "heuristic.h"

#define HEUR_FN __device__
#define HEUR_SHARED __shared__
#define HEUR_TID threadIdx.x
#define HEUR_NT blockDim.x
#define HEUR_SYNC() __syncthreads()
#ifdef WHAMD_HEURISTIC_STAMPS
#define HEUR_STAMP_BEGIN(D) do { if (threadIdx.x == 0) (D).stats[7] = __builtin_readcyclecounter(); } while (0)
#define HEUR_STAMP(D, phase) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); (D).stats[8 + (phase)] += now_ - (D).stats[7]; (D).stats[7] = now_; } } while (0)
#endif
namespace whamd {
__device__ __forceinline__ uint32_t heur_cas32(uint32_t* p, uint32_t cmp, uint32_t val) { return atomicCAS(p, cmp, val); }
__device__ __forceinline__ void heur_min32(uint32_t* p, uint32_t v) { atomicMin(p, v); }
__device__ __forceinline__ void heur_min64(unsigned long long* p, unsigned long long v) { atomicMin(p, v); }
__device__ __forceinline__ uint32_t heur_add32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
// inclusive prefix sum over the block: shuffles inside a wavefront, one pass over the wave totals (three barriers instead of 2 log n)
__device__ __forceinline__ uint32_t heur_block_inclusive(uint32_t v, uint32_t* tmp) {
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63u) >> 6;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t u = (uint32_t)__shfl_up((int)v, off);
		if (lane >= (uint32_t)off) v += u;
	}
	if (lane == 63u) tmp[wave] = v;
	__syncthreads();
	if (wave == 0) {
		uint32_t t = lane < n_waves ? tmp[lane] : 0u;
#pragma unroll
		for (int off = 1; off < 16; off <<= 1) {
			const uint32_t u = (uint32_t)__shfl_up((int)t, off);
			if (lane >= (uint32_t)off) t += u;
		}
		if (lane < n_waves) tmp[lane] = t;
	}
	__syncthreads();
	if (wave > 0) v += tmp[wave - 1u];
	__syncthreads();
	return v;
}
__device__ __forceinline__ uint32_t heur_load32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long heur_load64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}  // namespace whamd
#include "heuristic_core.h"

namespace whamd {

namespace {
template <int MAXT>
__global__ __launch_bounds__(MAXT) void heuristic_kernel(HeurDev D) { heur_solve(D); }
}  // namespace

whamd_status_t heuristic_solve_device(const HeurPlan& pl, int device, HeurResult& out, std::string& msg) {
	out = HeurResult();
	out.bipartition.assign(pl.n_reads, 0);
	out.transmission.assign(pl.n_cols, 0);
	if (pl.n_cols == 0) return WHAMD_OK;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		msg = "no HIP device visible: the whatshap_amd device path needs an MI355X (gfx950); there is no CPU fallback";
		return WHAMD_ERR_DEVICE;
	}
	if (device < 0 || device >= ndev) { msg = "device index " + std::to_string(device) + " out of range (" + std::to_string(ndev) + " visible)"; return WHAMD_ERR_DEVICE; }
#define HEUR_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { msg = std::string(#expr) + " failed: " + hipGetErrorString(e_); for (void* a : allocations) (void)hipFree(a); return WHAMD_ERR_DEVICE; } } while (0)
	std::vector<void*> allocations;
	HEUR_TRY(hipSetDevice(device));
	size_t free_b = 0, total_b = 0;
	HEUR_TRY(hipMemGetInfo(&free_b, &total_b));
	const uint32_t T = 1u << pl.tm_bits;
	const size_t rows = 2u * pl.n_samples;
	// worst case of the beam: the pruning keeps every solution that ties with the best one, up to 65535 (filterSolutions), a read doubles
	// them, the transmission values multiply them by 4^trios.  Sized for that when it fits in a third of the free memory, else for
	// 16 x row_limit (an overflow is reported, never silently pruned differently).
	const size_t per_solution = rows * pl.w_max * 4 + (size_t)pl.nw * 8 + 48;
	uint64_t cap = (uint64_t)HEUR_MAX_ROW_LIMIT * std::max(2u, T);
	if (2 * cap * per_solution > free_b / 3) cap = std::max<uint64_t>((uint64_t)pl.row_limit * 16u * std::max(2u, T), 4096);
	if (2 * cap * per_solution > free_b / 3) { msg = "PedMecHeuristic: the solution pools do not fit in device memory"; return WHAMD_ERR_UNSUPPORTED; }
	uint32_t tsz = 64;
	while (tsz < 2 * cap) tsz <<= 1;
	auto alloc = [&](void** dptr, size_t bytes) -> hipError_t {
		hipError_t e = hipMalloc(dptr, std::max<size_t>(bytes, 16));
		if (e == hipSuccess) allocations.push_back(*dptr);
		return e;
	};
	auto up = [&](void** dptr, const void* src, size_t bytes) -> hipError_t {
		hipError_t e = alloc(dptr, bytes);
		if (e == hipSuccess && bytes) e = hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice);
		return e;
	};
	HeurDev D{};
	D.n_cols = pl.n_cols; D.n_samples = pl.n_samples; D.n_trios = pl.n_trios; D.tm_bits = pl.tm_bits; D.row_limit = pl.row_limit;
	D.distrust = pl.distrust; D.w_max = pl.w_max; D.nw = pl.nw;
	const std::vector<HeurColMeta> col_meta = heuristic_col_meta(pl);
	const std::vector<HeurReadMeta> read_meta = heuristic_read_meta(pl);
	void* d = nullptr;
#define HEUR_UP(field, vec, type) HEUR_TRY(up(&d, (vec).data(), (vec).size() * sizeof((vec)[0]))); D.field = (type)d
	HEUR_UP(trios, pl.trios, const uint32_t*);
	HEUR_UP(recomb, pl.recomb, const float*); HEUR_UP(mutation, pl.mutation, const float*); HEUR_UP(genotype, pl.genotype, const int8_t*);
	HEUR_UP(start_index, pl.start_index, const uint32_t*); HEUR_UP(col, col_meta, const HeurColMeta*); HEUR_UP(kept, pl.kept, const uint32_t*);
	HEUR_UP(reads, read_meta, const HeurReadMeta*); HEUR_UP(new_balance, pl.new_balance, const float*); HEUR_UP(new_target, pl.new_target, const int32_t*);
#undef HEUR_UP
	D.cap = (uint32_t)cap; D.tsz = tsz;
	for (int q = 0; q < 2; ++q) HEUR_TRY(alloc((void**)&D.pool_words[q], heur_pool_words(D.cap, pl.nw, pl.n_samples, pl.w_max) * 4));
	HEUR_TRY(alloc((void**)&D.scratch, heur_scratch_words(D.cap, pl.nw) * 4));
	HEUR_TRY(alloc((void**)&D.hash, heur_hash_words(tsz) * 4));
	HEUR_TRY(alloc((void**)&D.col_off, (size_t)pl.n_cols * 8)); HEUR_TRY(alloc((void**)&D.col_count, (size_t)pl.n_cols * 4));
	HEUR_TRY(alloc((void**)&D.opt_bipart, std::max<size_t>(pl.n_reads, 1))); HEUR_TRY(alloc((void**)&D.opt_trans, (size_t)pl.n_cols * 4));
	HEUR_TRY(alloc((void**)&D.stats, 256));
	HEUR_TRY(hipMemset(D.opt_bipart, 0, std::max<size_t>(pl.n_reads, 1)));
	hipEvent_t ev0, ev1;
	HEUR_TRY(hipEventCreate(&ev0));
	HEUR_TRY(hipEventCreate(&ev1));
	// the backtrace arena: sized for 4 x row_limit x 4^trios solutions per column first, regrown on overflow while memory allows
	unsigned long long stride_sum = 0;
	for (uint32_t p = 0; p < pl.n_cols; ++p) stride_sum += 2 + ((pl.n_new[p] + 31) >> 5);
	unsigned long long arena_words = stride_sum * std::min<uint64_t>(cap, (uint64_t)pl.row_limit * 4u * T) + 1024;
	unsigned long long stats[32] = {0};
	whamd_status_t status = WHAMD_OK;
	for (;;) {
		HEUR_TRY(hipMemGetInfo(&free_b, &total_b));
		if (arena_words * 4 > free_b / 2) { msg = "PedMecHeuristic: the backtrace records do not fit in device memory"; status = WHAMD_ERR_UNSUPPORTED; break; }
		void* arena = nullptr;
		HEUR_TRY(hipMalloc(&arena, arena_words * 4));
		D.arena = (uint32_t*)arena; D.arena_words = arena_words;
		HEUR_TRY(hipMemset(D.stats, 0, 256));
		HEUR_TRY(hipEventRecord(ev0, nullptr));
		// as many threads as the beam usually has solutions (a barrier costs with the number of waves): 2 x row_limit, 128 .. 1024
		uint32_t block = 128;
		while (block < 1024u && block < 2u * pl.row_limit) block <<= 1;
		if (const char* e = getenv("WHAMD_HEURISTIC_THREADS")) block = (uint32_t)std::max(64, std::min(1024, atoi(e)));
		// (compiled twice: up to 512 threads a thread may hold 256 VGPRs -- at the 128 of a 1024-thread workgroup the batches of the row
		// copies spill to scratch memory)
		if (block <= 512u) hipLaunchKernelGGL(heuristic_kernel<512>, dim3(1), dim3(block), 0, nullptr, D);
		else hipLaunchKernelGGL(heuristic_kernel<1024>, dim3(1), dim3(block), 0, nullptr, D);
		HEUR_TRY(hipEventRecord(ev1, nullptr));
		hipError_t e = hipDeviceSynchronize();
		if (e == hipSuccess) e = hipMemcpy(stats, D.stats, sizeof stats, hipMemcpyDeviceToHost);
		(void)hipFree(arena);
		if (e != hipSuccess) { msg = std::string("heuristic kernel failed: ") + hipGetErrorString(e); status = WHAMD_ERR_DEVICE; break; }
		if (stats[0] == 2) { arena_words *= 4; continue; }
		if (stats[0] == 1) { msg = "PedMecHeuristic: more tied solutions than the device pools hold (" + std::to_string(cap) + ")"; status = WHAMD_ERR_UNSUPPORTED; }
		break;
	}
#ifdef WHAMD_HEURISTIC_STAMPS
	fprintf(stderr, "[whamd heuristic stamps] cycles: hash %llu, project-copy %llu, read pass 1 %llu, read pass 2 %llu, filter %llu, transmissions %llu, phasing + record %llu; hash parts: init %llu, gather %llu\n",
	        stats[8], stats[9], stats[10], stats[11], stats[12], stats[13], stats[14], stats[15], stats[16]);
#endif
	if (status == WHAMD_OK) {
		float ms = 0;
		(void)hipEventElapsedTime(&ms, ev0, ev1);
		out.device_ms = ms;
		out.max_solutions = stats[1];
		out.total_solutions = stats[2];
		hipError_t e = hipMemcpy(out.transmission.data(), D.opt_trans, (size_t)pl.n_cols * 4, hipMemcpyDeviceToHost);
		if (e == hipSuccess && pl.n_reads) e = hipMemcpy(out.bipartition.data(), D.opt_bipart, pl.n_reads, hipMemcpyDeviceToHost);
		if (e != hipSuccess) { msg = std::string("download failed: ") + hipGetErrorString(e); status = WHAMD_ERR_DEVICE; }
	}
	(void)hipEventDestroy(ev0);
	(void)hipEventDestroy(ev1);
	for (void* a : allocations) (void)hipFree(a);
#undef HEUR_TRY
	return status;
}

}  // namespace whamd
