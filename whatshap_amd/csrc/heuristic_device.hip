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
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "device_pool.h"
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

// One launch = SEVERAL tables, one persistent workgroup each (blockIdx.x selects the table's descriptor): independent chromosomes /
// families run side by side on as many CUs -- the beam of one table cannot use more than one.
template <int MAXT>
__global__ __launch_bounds__(MAXT) void heuristic_kernel_many(const HeurDev* __restrict__ tables) {
	HeurDev D;   // (copied through the constant address space: wave-uniform scalar loads, the descriptor ends up in SGPRs like a kernel argument)
	{
		static_assert(sizeof(HeurDev) % 4 == 0, "whole words");
		uint32_t* w = reinterpret_cast<uint32_t*>(&D);
		const __attribute__((address_space(4))) uint32_t* src = (const __attribute__((address_space(4))) uint32_t*)(unsigned long long)(tables + blockIdx.x);
#pragma unroll
		for (uint32_t i = 0; i < sizeof(HeurDev) / 4; ++i) w[i] = src[i];
	}
	heur_solve(D);
}

}  // namespace

void heuristic_release_cache() { devpool_release(); }

// Several tables in flight: everything between "plans built" and "bipartitions on the host".
struct HeurBatch::Impl {
	int device = 0;
	hipStream_t stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	struct Job {
		const HeurPlan* plan = nullptr;
		HeurDev D{};
		std::vector<std::pair<void*, size_t>> blocks;   // pool blocks (pointer, class size)
		void* arena = nullptr; size_t arena_bytes = 0;
		unsigned long long arena_words = 0;
		uint64_t cap = 0;
		uint32_t block = 128;
		bool done = false;
		unsigned long long stats[32] = {0};
	};
	std::vector<Job> jobs;
	HeurDev* d_tables = nullptr; size_t d_tables_bytes = 0;
	bool launched = false;
	~Impl() {
		(void)hipSetDevice(device);
		if (stream) (void)hipStreamSynchronize(stream);
		for (Job& j : jobs) {
			for (auto& b : j.blocks) devpool_give(device, b.first, b.second);
			devpool_give(device, j.arena, j.arena_bytes);
		}
		devpool_give(device, d_tables, d_tables_bytes);
		if (ev0) (void)hipEventDestroy(ev0);
		if (ev1) (void)hipEventDestroy(ev1);
		if (stream) (void)hipStreamDestroy(stream);
	}
};

HeurBatch::HeurBatch() : impl_(new Impl()) {}
HeurBatch::~HeurBatch() { delete impl_; }

#define HEUR_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { msg = std::string(#expr) + " failed: " + hipGetErrorString(e_); return WHAMD_ERR_DEVICE; } } while (0)

// Launches every job that is not done yet: one launch per workgroup size (the kernel is compiled for 512 and for 1024 threads).
static whamd_status_t heur_launch_pending(HeurBatch::Impl& m, std::string& msg) {
	std::vector<HeurDev> host;
	std::vector<size_t> which;
	size_t written = 0;   // descriptors of this round's launches: consecutive parts of d_tables (a later launch must not overwrite an earlier one's)
	for (uint32_t block : {128u, 256u, 512u, 1024u}) {
		host.clear();
		which.clear();
		for (size_t i = 0; i < m.jobs.size(); ++i)
			if (!m.jobs[i].done && m.jobs[i].block == block) { host.push_back(m.jobs[i].D); which.push_back(i); }
		if (host.empty()) continue;
		if (host.size() == 1) {   // a single table: descriptor by value (kernel arguments)
			if (block <= 512u) hipLaunchKernelGGL(heuristic_kernel<512>, dim3(1), dim3(block), 0, m.stream, host[0]);
			else hipLaunchKernelGGL(heuristic_kernel<1024>, dim3(1), dim3(block), 0, m.stream, host[0]);
			continue;
		}
		HeurDev* dst = m.d_tables + written;
		written += host.size();
		HEUR_TRY(hipMemcpyAsync(dst, host.data(), host.size() * sizeof(HeurDev), hipMemcpyHostToDevice, m.stream));
		HEUR_TRY(hipStreamSynchronize(m.stream));   // (`host` is pageable and dies with this scope)
		if (block <= 512u) hipLaunchKernelGGL(heuristic_kernel_many<512>, dim3((uint32_t)host.size()), dim3(block), 0, m.stream, dst);
		else hipLaunchKernelGGL(heuristic_kernel_many<1024>, dim3((uint32_t)host.size()), dim3(block), 0, m.stream, dst);
	}
	HEUR_TRY(hipGetLastError());
	return WHAMD_OK;
}

whamd_status_t HeurBatch::enqueue(const HeurPlan* const* plans, size_t n, int device, std::string& msg) {
	Impl& m = *impl_;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		msg = "no HIP device visible: the whatshap_amd device path needs an MI355X (gfx950); there is no CPU fallback";
		return WHAMD_ERR_DEVICE;
	}
	if (device < 0 || device >= ndev) { msg = "device index " + std::to_string(device) + " out of range (" + std::to_string(ndev) + " visible)"; return WHAMD_ERR_DEVICE; }
	m.device = device;
	HEUR_TRY(hipSetDevice(device));
	HEUR_TRY(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
	HEUR_TRY(hipEventCreate(&m.ev0));
	HEUR_TRY(hipEventCreate(&m.ev1));
	size_t free_b = 0, total_b = 0;
	HEUR_TRY(hipMemGetInfo(&free_b, &total_b));
	const size_t budget = free_b / 3 / std::max<size_t>(n, 1);   // what one table's pools may take
	m.jobs.resize(n);
	{
		size_t got = 0;
		void* ptr = nullptr;
		HEUR_TRY(devpool_take(device, (n + 1) * sizeof(HeurDev), &ptr, &got));
		m.d_tables = (HeurDev*)ptr; m.d_tables_bytes = got;
	}
	for (size_t ji = 0; ji < n; ++ji) {
		const HeurPlan& pl = *plans[ji];
		Impl::Job& job = m.jobs[ji];
		job.plan = &pl;
		if (pl.n_cols == 0) { job.done = true; continue; }
		const uint32_t T = 1u << pl.tm_bits;
		const size_t rows = 2u * pl.n_samples;
		// worst case of the beam: the pruning keeps every solution that ties with the best one, up to 65535 (filterSolutions), a read doubles
		// them, the transmission values multiply them by 4^trios.  Sized for that when it fits in the table's share of the free memory, else
		// for 16 x row_limit (an overflow is reported, never silently pruned differently).
		const size_t per_solution = rows * pl.w_max * 4 + (size_t)pl.nw * 8 + 48;
		const uint64_t full_cap = (uint64_t)HEUR_MAX_ROW_LIMIT * std::max(2u, T);
		uint64_t cap = full_cap;
		if (2 * cap * per_solution > budget) cap = std::min<uint64_t>(full_cap, std::max<uint64_t>((uint64_t)pl.row_limit * 16u * std::max(2u, T), 4096));
		if (2 * cap * per_solution > budget) { msg = "PedMecHeuristic: the solution pools do not fit in device memory"; return WHAMD_ERR_UNSUPPORTED; }
		job.cap = cap;
		uint32_t tsz = 64;
		while (tsz < 2 * cap) tsz <<= 1;
		auto alloc = [&](void** dptr, size_t bytes) -> hipError_t {
			size_t got = 0;
			hipError_t e = devpool_take(device, std::max<size_t>(bytes, 16), dptr, &got);
			if (e == hipSuccess) job.blocks.emplace_back(*dptr, got);
			return e;
		};
		// the small read-only arrays of the plan travel as ONE block (one allocation, one copy)
		HeurDev& D = job.D;
		D.n_cols = pl.n_cols; D.n_samples = pl.n_samples; D.n_trios = pl.n_trios; D.tm_bits = pl.tm_bits; D.row_limit = pl.row_limit;
		D.distrust = pl.distrust; D.w_max = pl.w_max; D.nw = pl.nw;
		const std::vector<HeurColMeta> col_meta = heuristic_col_meta(pl);
		const std::vector<HeurReadMeta> read_meta = heuristic_read_meta(pl);
		struct Piece { const void* src; size_t bytes; size_t off; };
		std::vector<Piece> pieces;
		size_t total = 0;
		auto piece = [&](const void* src, size_t bytes) { const size_t off = total; pieces.push_back(Piece{src, bytes, off}); total += (bytes + 255) & ~(size_t)255; return off; };
		const size_t o_trios = piece(pl.trios.data(), pl.trios.size() * 4), o_recomb = piece(pl.recomb.data(), pl.recomb.size() * 4);
		const size_t o_mut = piece(pl.mutation.data(), pl.mutation.size() * 4), o_geno = piece(pl.genotype.data(), pl.genotype.size());
		const size_t o_start = piece(pl.start_index.data(), pl.start_index.size() * 4), o_col = piece(col_meta.data(), col_meta.size() * sizeof(HeurColMeta));
		const size_t o_kept = piece(pl.kept.data(), pl.kept.size() * 4), o_reads = piece(read_meta.data(), read_meta.size() * sizeof(HeurReadMeta));
		const size_t o_bal = piece(pl.new_balance.data(), pl.new_balance.size() * 4), o_target = piece(pl.new_target.data(), pl.new_target.size() * 4);
		std::vector<char> staged(std::max<size_t>(total, 16), 0);
		for (const Piece& pc : pieces) if (pc.bytes) std::memcpy(staged.data() + pc.off, pc.src, pc.bytes);
		char* base = nullptr;
		HEUR_TRY(alloc((void**)&base, staged.size()));
		HEUR_TRY(hipMemcpyAsync(base, staged.data(), staged.size(), hipMemcpyHostToDevice, m.stream));
		HEUR_TRY(hipStreamSynchronize(m.stream));   // (`staged` is pageable)
		D.trios = (const uint32_t*)(base + o_trios); D.recomb = (const float*)(base + o_recomb); D.mutation = (const float*)(base + o_mut);
		D.genotype = (const int8_t*)(base + o_geno); D.start_index = (const uint32_t*)(base + o_start); D.col = (const HeurColMeta*)(base + o_col);
		D.kept = (const uint32_t*)(base + o_kept); D.reads = (const HeurReadMeta*)(base + o_reads); D.new_balance = (const float*)(base + o_bal);
		D.new_target = (const int32_t*)(base + o_target);
		D.cap = (uint32_t)cap; D.tsz = tsz;
		for (int q = 0; q < 2; ++q) HEUR_TRY(alloc((void**)&D.pool_words[q], heur_pool_words(D.cap, pl.nw, pl.n_samples, pl.w_max) * 4));
		HEUR_TRY(alloc((void**)&D.scratch, heur_scratch_words(D.cap, pl.nw) * 4));
		HEUR_TRY(alloc((void**)&D.hash, heur_hash_words(tsz) * 4));
		// col_off | col_count | opt_trans | opt_bipart | stats: one block
		const size_t b_off = 0, b_count = (size_t)pl.n_cols * 8, b_trans = b_count + (((size_t)pl.n_cols * 4 + 7) & ~(size_t)7), b_bip = b_trans + (((size_t)pl.n_cols * 4 + 7) & ~(size_t)7);
		const size_t b_stats = b_bip + ((std::max<size_t>(pl.n_reads, 1) + 255) & ~(size_t)255), b_total = b_stats + 256;
		char* res = nullptr;
		HEUR_TRY(alloc((void**)&res, b_total));
		D.col_off = (unsigned long long*)(res + b_off); D.col_count = (uint32_t*)(res + b_count); D.opt_trans = (uint32_t*)(res + b_trans);
		D.opt_bipart = (uint8_t*)(res + b_bip); D.stats = (unsigned long long*)(res + b_stats);
		HEUR_TRY(hipMemsetAsync(res + b_bip, 0, b_total - b_bip, m.stream));
		// the backtrace arena: sized for 4 x row_limit x 4^trios solutions per column first, regrown on overflow while memory allows
		unsigned long long stride_sum = 0;
		for (uint32_t p = 0; p < pl.n_cols; ++p) stride_sum += 2 + ((pl.n_new[p] + 31) >> 5);
		job.arena_words = stride_sum * std::min<uint64_t>(cap, (uint64_t)pl.row_limit * 4u * T) + 1024;
		if (job.arena_words * 4 > free_b / 2 / std::max<size_t>(n, 1)) { msg = "PedMecHeuristic: the backtrace records do not fit in device memory"; return WHAMD_ERR_UNSUPPORTED; }
		HEUR_TRY(devpool_take(device, job.arena_words * 4, &job.arena, &job.arena_bytes));
		D.arena = (uint32_t*)job.arena; D.arena_words = job.arena_words;
		// as many threads as the beam usually has solutions (a barrier costs with the number of waves): 2 x row_limit, 128 .. 1024
		uint32_t block = 128;
		while (block < 1024u && block < 2u * pl.row_limit) block <<= 1;
		if (const char* e = getenv("WHAMD_HEURISTIC_THREADS")) { block = 128; const uint32_t want = (uint32_t)std::max(64, std::min(1024, atoi(e))); while (block < want) block <<= 1; }
		job.block = block;
	}
	HEUR_TRY(hipEventRecord(m.ev0, m.stream));
	const whamd_status_t st = heur_launch_pending(m, msg);
	if (st != WHAMD_OK) return st;
	HEUR_TRY(hipEventRecord(m.ev1, m.stream));
	m.launched = true;
	return WHAMD_OK;
}

whamd_status_t HeurBatch::wait(HeurResult* outs, std::string& msg) {
	Impl& m = *impl_;
	if (!m.launched) { msg = "the batch was not enqueued"; return WHAMD_ERR_INVALID; }
	HEUR_TRY(hipSetDevice(m.device));
	float ms_total = 0;
	for (;;) {
		hipError_t e = hipStreamSynchronize(m.stream);
		if (e != hipSuccess) { msg = std::string("heuristic kernel failed: ") + hipGetErrorString(e); return WHAMD_ERR_DEVICE; }
		float ms = 0;
		(void)hipEventElapsedTime(&ms, m.ev0, m.ev1);
		ms_total += ms;
		bool again = false;
		for (Impl::Job& job : m.jobs) {
			if (job.done) continue;
			HEUR_TRY(hipMemcpy(job.stats, job.D.stats, sizeof job.stats, hipMemcpyDeviceToHost));
			if (job.stats[0] == 2) {   // the records outgrew the arena: four times the room, this table once more
				devpool_give(m.device, job.arena, job.arena_bytes);
				job.arena = nullptr;
				job.arena_words *= 4;
				size_t free_b = 0, total_b = 0;
				HEUR_TRY(hipMemGetInfo(&free_b, &total_b));
				if (job.arena_words * 4 > free_b / 2) { msg = "PedMecHeuristic: the backtrace records do not fit in device memory"; return WHAMD_ERR_UNSUPPORTED; }
				HEUR_TRY(devpool_take(m.device, job.arena_words * 4, &job.arena, &job.arena_bytes));
				job.D.arena = (uint32_t*)job.arena; job.D.arena_words = job.arena_words;
				HEUR_TRY(hipMemsetAsync(job.D.stats, 0, 256, m.stream));
				HEUR_TRY(hipMemsetAsync(job.D.opt_bipart, 0, std::max<size_t>(job.plan->n_reads, 1), m.stream));
				again = true;
				continue;
			}
			if (job.stats[0] == 1) { msg = "PedMecHeuristic: more tied solutions than the device pools hold (" + std::to_string(job.cap) + ")"; return WHAMD_ERR_UNSUPPORTED; }
			job.done = true;
		}
		if (!again) break;
		HEUR_TRY(hipEventRecord(m.ev0, m.stream));
		const whamd_status_t st = heur_launch_pending(m, msg);
		if (st != WHAMD_OK) return st;
		HEUR_TRY(hipEventRecord(m.ev1, m.stream));
	}
#ifdef WHAMD_HEURISTIC_STAMPS
	for (const Impl::Job& job : m.jobs)
		fprintf(stderr, "[whamd heuristic stamps] cycles: hash %llu, project-copy %llu, read pass 1 %llu, read pass 2 %llu, filter %llu, transmissions %llu, phasing + record %llu; hash parts: init %llu, gather %llu\n",
		        job.stats[8], job.stats[9], job.stats[10], job.stats[11], job.stats[12], job.stats[13], job.stats[14], job.stats[15], job.stats[16]);
#endif
	for (size_t ji = 0; ji < m.jobs.size(); ++ji) {
		const Impl::Job& job = m.jobs[ji];
		const HeurPlan& pl = *job.plan;
		HeurResult& out = outs[ji];
		out = HeurResult();
		out.bipartition.assign(pl.n_reads, 0);
		out.transmission.assign(pl.n_cols, 0);
		if (pl.n_cols == 0) continue;
		out.device_ms = ms_total;   // (the launch all tables of the batch shared)
		out.max_solutions = job.stats[1];
		out.total_solutions = job.stats[2];
		HEUR_TRY(hipMemcpy(out.transmission.data(), job.D.opt_trans, (size_t)pl.n_cols * 4, hipMemcpyDeviceToHost));
		if (pl.n_reads) HEUR_TRY(hipMemcpy(out.bipartition.data(), job.D.opt_bipart, pl.n_reads, hipMemcpyDeviceToHost));
	}
	return WHAMD_OK;
}
#undef HEUR_TRY

whamd_status_t heuristic_solve_device(const HeurPlan& pl, int device, HeurResult& out, std::string& msg) {
	HeurBatch batch;
	const HeurPlan* plans[1] = {&pl};
	whamd_status_t st = batch.enqueue(plans, 1, device, msg);
	if (st != WHAMD_OK) return st;
	return batch.wait(&out, msg);
}

}  // namespace whamd
