// dp_device.hip -- gfx950 kernels and the device driver of the wMEC / PedMEC forward pass + backtrace.
//
// What is computed (bit-exact restatement of src/pedigreedptable.cpp:177-335):
//   D_c[x][i]  = cost_{c,i}(x) (+) min_j ( Pr_{c-1}[x & lowmask_b][j] + popcount(i^j) * recomb_c ),  lowest j on ties
//   Pr_c[y][i] = min { D_c[x][i] : pext(x, fwd_mask_c) == y },  argmin = the x with the smallest Gray-code rank
// The reference walks x in reflected-Gray-code order with strict '<' updates; here every cell is evaluated
// independently (closed-form cost, no Gray stepping) and ties are broken with the key (value, gray_rank(x)).
//
// Kernels (included below, one file per family):
//   kernels_column.h     one launch per column: column_step_fused (thread = one projection entry, ballot-packed argmin
//                        planes), column_step_keys + column_finalize (64-bit atomicMin keys; many ending reads, tiny
//                        columns, the last column)
//   kernels_resident.h   single-individual runs: ~23 columns per launch, the projection column lives in LDS
//   kernels_trio.h       trio runs (T = 4)
//   kernels_backtrace.h  follows the stored argmins from the last column to the first (src/pedigreedptable.cpp:137-173)
// This file: launch tables and the host driver (DeviceTable: upload, jobs / lanes, resumable submission, wait).
// No MFMA (integer min-plus), no CUDA compatibility layer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "debug_build.h"
#include "device_pool.h"
#include "device_table.h"
#include "genotype.h"

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

// device code, in dependency order (each file documents its kernels)
#include "kernels_column.h"
#include "kernels_resident.h"
#include "kernels_trio.h"
#include "kernels_slots.h"
#include "kernels_pedslots.h"
#include "kernels_backtrace.h"

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

// ---------------------------------------------------------------------------------------------- pinned staging of the create path
// hipMemcpyAsync from pageable memory goes through the runtime's own small staging buffers: ~5 GB/s measured for the ~100 MB of
// descriptors of a configs[2] table, 20 ms of a 58 ms create.  The create path copies its large arrays into ONE process-wide pinned
// area with a few host threads and sends them from there (the copies overlap the host work that follows).  One upload() at a
// time owns the area; it grows to what the largest table so far needed (at most STAGE_MAX; larger uploads go in rounds).
// ---------------------------------------------------------------------------------------------- device_pool.h
struct DevPool {
	std::mutex mu;
	struct Block { void* ptr; size_t bytes; int device; };
	std::vector<Block> idle;
	size_t idle_bytes = 0;
};
DevPool g_pool;
constexpr size_t POOL_KEEP = (size_t)24 << 30;   // most bytes kept idle
size_t pool_class(size_t bytes) {
	bytes = std::max<size_t>(bytes, 256);
	size_t p = 256;
	while (p < bytes) p <<= 1;
	const size_t step = std::max<size_t>(p / 8, 256);
	return (bytes + step - 1) / step * step;
}

// A table's stream and events, and its pinned download buffer, come from pools as well: creating and destroying them per table (a stream,
// five events, hipHostMalloc / hipHostFree of the path buffer) was as expensive as the whole create of a coverage-15 table
// (24 tables: create 103 ms on 8 threads, close 110 ms).
struct StreamSet { hipStream_t stream = nullptr; hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; int device = -1; };
struct HostBlock { void* ptr; size_t bytes; };
struct MiscPool {
	std::mutex mu;
	std::vector<StreamSet> streams;
	std::vector<HostBlock> pinned;
	size_t pinned_bytes = 0;
};
MiscPool g_misc;
bool streamset_take(int device, StreamSet& out) {
	{
		std::lock_guard<std::mutex> lock(g_misc.mu);
		for (size_t i = 0; i < g_misc.streams.size(); ++i) {
			if (g_misc.streams[i].device != device) continue;
			out = g_misc.streams[i];
			g_misc.streams[i] = g_misc.streams.back();
			g_misc.streams.pop_back();
			return true;
		}
	}
	out = StreamSet();
	out.device = device;
	if (hipStreamCreateWithFlags(&out.stream, hipStreamNonBlocking) != hipSuccess) return false;
	for (int k = 0; k < 6; ++k)
		if ((k < 4 ? hipEventCreate(&out.ev[k]) : hipEventCreateWithFlags(&out.ev[k], hipEventDisableTiming)) != hipSuccess) return false;
	return true;
}
void streamset_give(const StreamSet& ss) {   // (the stream is idle: the caller synchronised it)
	if (!ss.stream) return;
	std::lock_guard<std::mutex> lock(g_misc.mu);
	if (g_misc.streams.size() < 256) { g_misc.streams.push_back(ss); return; }
	for (hipEvent_t e : ss.ev) if (e) (void)hipEventDestroy(e);
	(void)hipStreamDestroy(ss.stream);
}
// The streams the UPLOADS of all tables of a device go through: two, shared, used for nothing else.  A table's own stream carries its solve; an upload that went
// through it completed, under a running group solve, only when that solve's queue had drained: the staging areas came back late and 96 creates under a running solve
// took 137 - 190 ms instead of 81 - 93 ms alone (scripts/gpu_create_under_solve.py); on streams of their own: 72 - 101 ms.
struct UploadStreams {
	std::mutex mu;
	std::vector<std::pair<int, hipStream_t>> streams;   // (device, stream); never destroyed: they live as long as the process
	std::atomic<uint32_t> next{0};
};
UploadStreams g_upload_streams;
constexpr uint32_t UPLOAD_STREAMS = 2;   // (creating one costs ~10 ms, paid by the first creates of a process; the link serialises the copies anyway)
hipStream_t upload_stream_of(int device) {
	const uint32_t slot = g_upload_streams.next.fetch_add(1, std::memory_order_relaxed) % UPLOAD_STREAMS;
	std::lock_guard<std::mutex> lock(g_upload_streams.mu);
	uint32_t seen = 0;
	for (const auto& e : g_upload_streams.streams)
		if (e.first == device && seen++ == slot) return e.second;
	hipStream_t made = nullptr;
	while (seen <= slot) {
		int least = 0, greatest = 0;
		(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
		hipStream_t st = nullptr;
		// (default priority.  Streams of the HIGHEST priority -- debug library, WHAMD_UPLOAD_STREAMS_HIGH=1 -- were the first version: the creates under a running solve
		//  gained the same, but the mere existence of such streams made three full-width tables solved at once on their own streams 3.2 x slower, 177 ms against 55:
		//  scripts/gpu_wide_tables_concurrent.py)
		const bool high = debug_env("WHAMD_UPLOAD_STREAMS_HIGH") != nullptr;
		if ((high ? hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
		g_upload_streams.streams.emplace_back(device, st);
		made = st;
		++seen;
	}
	return made;
}
hipError_t pinned_take(size_t bytes, void** out, size_t* got) {
	const size_t want = pool_class(bytes);
	*got = want;
	{
		std::lock_guard<std::mutex> lock(g_misc.mu);
		for (size_t i = 0; i < g_misc.pinned.size(); ++i) {
			if (g_misc.pinned[i].bytes != want) continue;
			*out = g_misc.pinned[i].ptr;
			g_misc.pinned_bytes -= want;
			g_misc.pinned[i] = g_misc.pinned.back();
			g_misc.pinned.pop_back();
			return hipSuccess;
		}
	}
	return hipHostMalloc(out, want, hipHostMallocPortable);
}
void pinned_give(void* ptr, size_t bytes) {
	if (!ptr) return;
	{
		std::lock_guard<std::mutex> lock(g_misc.mu);
		if (g_misc.pinned_bytes + bytes <= ((size_t)1 << 30)) { g_misc.pinned.push_back(HostBlock{ptr, bytes}); g_misc.pinned_bytes += bytes; return; }
	}
	(void)hipHostFree(ptr);
}
void misc_pool_release() {
	std::vector<StreamSet> streams;
	std::vector<HostBlock> pinned;
	{
		std::lock_guard<std::mutex> lock(g_misc.mu);
		streams.swap(g_misc.streams);
		pinned.swap(g_misc.pinned);
		g_misc.pinned_bytes = 0;
	}
	int cur = 0;
	(void)hipGetDevice(&cur);
	for (const StreamSet& ss : streams) {
		(void)hipSetDevice(ss.device);
		for (hipEvent_t e : ss.ev) if (e) (void)hipEventDestroy(e);
		(void)hipStreamDestroy(ss.stream);
	}
	(void)hipSetDevice(cur);
	for (const HostBlock& b : pinned) (void)hipHostFree(b.ptr);
}

struct UploadStage {
	std::mutex mu;
	struct Area { char* base = nullptr; size_t cap = 0; bool busy = false; hipEvent_t ev = nullptr; bool parked = false; };   // parked: the last session left copies in flight, `ev` says when they are done
	std::vector<Area> areas;   // a few pinned areas: tables created by several host threads at once (blocks.solve_blocks) do not wait for each other
	size_t want = 0;
	bool broken = false;   // hipHostMalloc failed once: pageable copies from then on
};
UploadStage g_stage;
constexpr size_t STAGE_MAX = (size_t)1 << 30, STAGE_MIN_COPY = (size_t)256 << 10, STAGE_GRAIN = (size_t)16 << 20, STAGE_AREAS = 32;   // (eight areas: the ninth and later of 64 concurrent creates fell back to pageable copies, 9.5 GB/s and synchronous)

// ---------------------------------------------------------------------------------------------- arenas kept between tables
// hipFree + hipMalloc of a 13 GB backtrace arena per table stalls for up to a second every few tables (measured: create 26 ms,
// 26 ms, 26 ms, 997 ms; 24 tables of 100 000 columns created and released one after the other: 50 - 100 ms each).  The arenas of closed
// tables stay allocated (a few blocks per process, at most 60 % of the device); the next table takes the smallest one that is large enough
// and not wastefully large.  Counted as free memory when a table sizes its arena; given back when memory is tight.
struct ArenaCache {
	std::mutex mu;
	struct Block { void* ptr; size_t bytes; int device; };
	std::vector<Block> blocks;
};
ArenaCache g_arena;
constexpr size_t ARENA_BLOCKS = 512;   // (32 until round 6: the 96 tables of one step kept 32 arenas and hipFree-d 64 -- 13 ms of their releases -- and the next step allocated them again; the bytes are bounded separately, arena_give)
size_t arena_idle_bytes(int device) {
	std::lock_guard<std::mutex> lock(g_arena.mu);
	size_t sum = 0;
	for (const ArenaCache::Block& b : g_arena.blocks) if (b.device == device) sum += b.bytes;
	return sum;
}
void arena_free_block(const ArenaCache::Block& b) {
	int cur = 0;
	(void)hipGetDevice(&cur);
	(void)hipSetDevice(b.device);
	(void)hipFree(b.ptr);
	(void)hipSetDevice(cur);
}
// nullptr: nothing suitable.  `make_room`: the caller is about to hipMalloc `need` bytes -- blocks that do not fit are freed first.
void* arena_take(int device, size_t need, size_t& got, bool make_room = true) {
	std::vector<ArenaCache::Block> drop;
	void* out = nullptr;
	{
		std::lock_guard<std::mutex> lock(g_arena.mu);
		size_t best = g_arena.blocks.size();
		for (size_t i = 0; i < g_arena.blocks.size(); ++i) {
			const ArenaCache::Block& b = g_arena.blocks[i];
			if (b.device != device || b.bytes < need || b.bytes > 2 * need + ((size_t)1 << 30)) continue;
			if (best == g_arena.blocks.size() || b.bytes < g_arena.blocks[best].bytes) best = i;
		}
		if (best != g_arena.blocks.size()) {
			out = g_arena.blocks[best].ptr;
			got = g_arena.blocks[best].bytes;
			g_arena.blocks[best] = g_arena.blocks.back();
			g_arena.blocks.pop_back();
		} else if (make_room) {
			// nothing fits: the allocation that follows must not fail because of idle blocks -- free them when they are needed
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + ((size_t)4 << 30)) drop.swap(g_arena.blocks);
		}
	}
	for (const ArenaCache::Block& b : drop) arena_free_block(b);
	return out;
}
void arena_give(int device, void* ptr, size_t bytes) {   // called with `device` current
	if (!ptr) return;
	{
		std::lock_guard<std::mutex> lock(g_arena.mu);
		size_t idle = 0, total_b = 0, free_b = 0;
		for (const ArenaCache::Block& b : g_arena.blocks) idle += b.bytes;
		const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && idle + bytes <= total_b / 5 * 3;   // (eight trio tables of 100 000 columns: 8 x 15 GB)
		if (room && g_arena.blocks.size() < ARENA_BLOCKS && bytes >= ((size_t)32 << 20) && debug_env("WHAMD_NO_ARENA_CACHE") == nullptr) {
			g_arena.blocks.push_back(ArenaCache::Block{ptr, bytes, device});
			return;
		}
	}
	(void)hipFree(ptr);
}

struct StageSession {
	hipStream_t stream;
	size_t used = 0, total = 0;
	bool pending = false;
	const bool enabled;
	int slot = -1;          // the area this session owns (g_stage.areas), -1: none (pageable copies)
	char* base = nullptr;
	size_t cap = 0;
	explicit StageSession(hipStream_t s) : stream(s), enabled(debug_env("WHAMD_NO_PINNED_STAGE") == nullptr) {
		std::lock_guard<std::mutex> lock(g_stage.mu);
		size_t best = g_stage.areas.size();
		for (size_t i = 0; i < g_stage.areas.size(); ++i)
			if (!g_stage.areas[i].busy && (best == g_stage.areas.size() || (g_stage.areas[best].parked && !g_stage.areas[i].parked) ||
			                               (g_stage.areas[best].parked == g_stage.areas[i].parked && g_stage.areas[i].cap > g_stage.areas[best].cap))) best = i;
		if (best == g_stage.areas.size() && g_stage.areas.size() < STAGE_AREAS) { g_stage.areas.emplace_back(); best = g_stage.areas.size() - 1; }
		hipEvent_t wait_for = nullptr;
		if (best != g_stage.areas.size()) {
			slot = (int)best;
			g_stage.areas[best].busy = true;
			base = g_stage.areas[best].base;
			cap = g_stage.areas[best].cap;
			if (g_stage.areas[best].parked) wait_for = g_stage.areas[best].ev;
			g_stage.areas[best].parked = false;
		}
		if (wait_for) (void)hipEventSynchronize(wait_for);   // (the previous table's copies out of this area: normally long done)
	}
	// The table's create returns without waiting for the copies (DeviceTable::upload): the area stays reserved -- for the NEXT session -- behind an event.
	bool park() {
		if (slot < 0 || !pending) return true;
		std::lock_guard<std::mutex> lock(g_stage.mu);
		UploadStage::Area& a = g_stage.areas[slot];
		if (!a.ev && hipEventCreateWithFlags(&a.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); a.ev = nullptr; return false; }
		if (hipEventRecord(a.ev, stream) != hipSuccess) { (void)hipGetLastError(); return false; }
		a.parked = true;
		pending = false;
		used = 0;
		return true;
	}
	~StageSession() {
		if (pending) (void)hipStreamSynchronize(stream);
		std::lock_guard<std::mutex> lock(g_stage.mu);
		g_stage.want = std::max(g_stage.want, std::min(total, STAGE_MAX));
		if (slot >= 0) {
			g_stage.areas[slot].base = base;
			g_stage.areas[slot].cap = cap;
			g_stage.areas[slot].busy = false;
		}
	}
	bool ensure(size_t bytes) {   // area empty (nothing pending): make it hold `bytes`, or everything the largest table so far staged
		if (slot < 0) return false;
		size_t want = 0;
		{
			std::lock_guard<std::mutex> lock(g_stage.mu);
			want = g_stage.want;
		}
		const size_t need = (std::max(std::min(bytes, STAGE_MAX), want) + STAGE_GRAIN - 1) / STAGE_GRAIN * STAGE_GRAIN;
		if (cap >= need) return true;
		if (base) (void)hipHostFree(base);
		base = nullptr;
		cap = 0;
		void* ptr = nullptr;
		if (hipHostMalloc(&ptr, need, hipHostMallocPortable) != hipSuccess) {
			(void)hipGetLastError();
			g_stage.broken = true;
			return false;
		}
		base = (char*)ptr;
		cap = need;
		return true;
	}
	hipError_t copy(void* dst, const void* src, size_t bytes) {
		if (!enabled || g_stage.broken || slot < 0 || bytes < STAGE_MIN_COPY || image) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
		total += (bytes + 255) & ~(size_t)255;
		size_t done = 0;
		while (done < bytes) {
			if (!pending && used == 0 && !ensure(bytes - done)) return hipMemcpyAsync((char*)dst + done, (const char*)src + done, bytes - done, hipMemcpyHostToDevice, stream);
			const size_t chunk = std::min(bytes - done, cap - used);
			if (chunk == 0) {   // area full: wait for what is in flight, start over
				hipError_t e = hipStreamSynchronize(stream);
				if (e != hipSuccess) return e;
				pending = false;
				used = 0;
				continue;
			}
			char* at = base + used;
			const char* from = (const char*)src + done;
			parallel_ranges(chunk, host_threads(chunk, (size_t)2 << 20), [&](uint64_t b0, uint64_t b1, uint32_t) { std::memcpy(at + b0, from + b0, b1 - b0); });
			hipError_t e = hipMemcpyAsync((char*)dst + done, at, chunk, hipMemcpyHostToDevice, stream);
			if (e != hipSuccess) return e;
			pending = true;
			used += (chunk + 255) & ~(size_t)255;
			done += chunk;
		}
		return hipSuccess;
	}
	// One image of everything a table uploads (DeviceTable::upload): the area holds `bytes` and copy() leaves it alone (pieces that do not fit the image go
	// out as copies of their own, straight from the caller's memory).
	bool image = false;
	bool begin_image(size_t bytes) {
		if (!enabled || g_stage.broken || slot < 0 || pending || used != 0) return false;
		image = ensure(bytes);
		return image;
	}
	void expect(size_t bytes) {   // before the first copy: one allocation
		std::lock_guard<std::mutex> lock(g_stage.mu);
		g_stage.want = std::max(g_stage.want, std::min(bytes, STAGE_MAX));
	}
	void finish() {   // after the caller synchronised the stream
		pending = false;
		used = 0;
	}
};

}  // namespace

hipError_t devpool_take(int device, size_t bytes, void** out, size_t* got) {
	const size_t want = pool_class(bytes);
	*got = want;
	{
		std::lock_guard<std::mutex> lock(g_pool.mu);
		for (size_t i = 0; i < g_pool.idle.size(); ++i) {
			if (g_pool.idle[i].device != device || g_pool.idle[i].bytes != want) continue;
			*out = g_pool.idle[i].ptr;
			g_pool.idle_bytes -= want;
			g_pool.idle[i] = g_pool.idle.back();
			g_pool.idle.pop_back();
			return hipSuccess;
		}
	}
	hipError_t e = hipMalloc(out, want);
	if (e != hipSuccess) {   // out of memory with idle blocks around: give them back and try once more
		(void)hipGetLastError();
		devpool_release();
		e = hipMalloc(out, want);
	}
	return e;
}
void devpool_give(int device, void* ptr, size_t bytes) {
	if (!ptr) return;
	{
		std::lock_guard<std::mutex> lock(g_pool.mu);
		if (g_pool.idle_bytes + bytes <= POOL_KEEP && bytes <= ((size_t)2 << 30)) {
			g_pool.idle.push_back(DevPool::Block{ptr, bytes, device});
			g_pool.idle_bytes += bytes;
			return;
		}
	}
	(void)hipFree(ptr);
}
void devpool_release() {
	std::vector<DevPool::Block> blocks;
	{
		std::lock_guard<std::mutex> lock(g_pool.mu);
		blocks.swap(g_pool.idle);
		g_pool.idle_bytes = 0;
	}
	int cur = 0;
	(void)hipGetDevice(&cur);
	for (const DevPool::Block& b : blocks) { (void)hipSetDevice(b.device); (void)hipFree(b.ptr); }
	(void)hipSetDevice(cur);
}

// ================================================================================================ DeviceTable

struct DeviceTable::Impl {
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t run_stream = nullptr;   // where the forward steps of the solve being submitted go: `stream`, or the stream of the group's first table (enqueue_group)
	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
	hipEvent_t ev_group = nullptr;      // group solve: "the forward pass of every table of the group is submitted up to here"
	hipEvent_t ev_upload = nullptr;     // "everything upload() put on `stream` -- the copies, the table kernels -- is done": a solve waits for it ON THE DEVICE (begin_solve)
	std::vector<std::pair<void*, size_t>> allocations;   // (pointer, size class) of device_pool.h
	void* d_arena = nullptr;    // the backtrace arena: not in `allocations`, handed to the arena cache on release
	size_t arena_bytes = 0;
	DevColumn* d_cols = nullptr;
	uint32_t* d_pr[2] = {nullptr, nullptr};
	uint32_t* d_path_index = nullptr;
	uint32_t* d_path_trans = nullptr;
	uint32_t* d_score = nullptr;
	BtUnit* d_units = nullptr;
	std::vector<BtUnit> units;
	size_t bt_lds = 0;
	RawVec<DevColumn> cols;
	ResidentPlan plan;
	DevProblem dp{};
	FusedFn fused = nullptr;
	KeysFn keysfn = nullptr;
	bool wide = false;   // column_step_wide instead of the templated per-column kernels
	size_t key_entries = 0;
	std::string path = "auto";
	int l_pref = 11;
	bool fold = true;
	int symmetry = 1;  // single individual: compute only half of a run, the rest is its mirror image (plan_forward)
	uint64_t bt_bytes = 0;
	uint64_t launches = 0;
	uint32_t group_tables = 1;  // tables that shared the forward launches of the solve in flight (enqueue_group)
	uint32_t max_grid_x = 1;    // widest launch of the schedule, in workgroups
	bool enqueue_open = false;  // resumable enqueue (enqueue_some)
	size_t h_pinned_bytes = 0;
	uint32_t* h_pinned = nullptr;  // [2 n + jobs + 3]: path index, path transmission, score of the final job, scores of the others, the chunked backtrace's three counters
	// A job is a sequence of forward steps with its own backtrace.  Job 0 ("final") is the last connected component (it
	// ends with the table's last column, whose optimum comes from the key scratch); every other job is one earlier
	// connected component: it starts from cost 0, its last column projects onto a single entry -- that value is added on
	// the host -- and its backtrace starts at entry 0.  Jobs are independent: they are spread over lanes (private
	// exchange buffers and key scratch) and the lanes advance in lockstep, one *super-step* at a time: the runs of a
	// super-step go out as ONE batched launch (resident_batch), per-column steps as launches of their own; one backtrace
	// launch at the end walks all jobs, one workgroup each.
	struct Job {
		std::vector<uint32_t> steps;  // indices into plan.steps, execution order
		uint32_t unit_off = 0, unit_count = 0;
		bool final = false;
	};
	struct Lane {
		uint32_t* d_pr[2] = {nullptr, nullptr};
		unsigned long long* d_keys = nullptr;  // atomic-min scratch of the per-column kernels
		std::vector<uint32_t> jobs;
	};
	struct Single {        // a per-column step inside a super-step
		uint32_t lane, step, flip;
		bool zero_prev;    // first step of its job: the entry it may read must hold cost 0
		int32_t score_job; // >= 0: last step of that (non-final) job: copy its single exit value aside
	};
	struct SuperStep {
		uint32_t entry_off = 0, entry_count = 0, grid_x = 0, threads = 0;
		size_t lds = 0;
		bool sym = false;  // some run of the batch takes part in the complement symmetry
		std::vector<Single> singles;
		// windowed solve: restore the kept column of boundary ck_load into this step's input before it runs; keep this
		// step's output as boundary ck_save; walk window bt_window after it
		int32_t ck_load = -1, ck_save = -1, bt_window = -1;
		uint32_t* io[2] = {nullptr, nullptr};   // input / output exchange buffer of the step (single lane)
	};
	std::vector<Job> jobs;
	std::vector<Lane> lanes;
	std::vector<SuperStep> schedule;
	// What a GROUP submission (enqueue_group) needs of a super-step and of its entries, a few bytes each: with 96 tables per launch the submitting thread read a
	// 320-byte SlotBatchEntry and a SuperStep per table and super-step out of cold memory -- 31 ms of host time for 2 276 super-steps, 13 ms with these
	// (scripts/gpu_group_submit_ab.py).  Built at create time (where the entries are made, on the create's own threads).
	struct StepBrief { uint32_t entry_off, lds; uint16_t entry_count; uint8_t has_singles, pad; };
	struct EntryBrief { uint16_t grid_x, threads; uint32_t lds_x; uint8_t variant, variant_x, pad[2]; };   // variant_x: the X kernel's variant where the run is eligible (else = variant)
	std::vector<StepBrief> step_brief;
	std::vector<EntryBrief> entry_brief;
	std::vector<ResBatchEntry> entries;
	ResBatchEntry* d_entries = nullptr;
	// chunked speculative backtrace (kernels_backtrace.h) of a table made of slot runs
	bool use_chunks = false;
	std::vector<BtChunk> chunks;
	BtChunk* d_chunks = nullptr;
	uint32_t* d_unit_x = nullptr;   // [2][units]
	uint32_t* d_path2 = nullptr;    // [2][columns]: speculative walks of the two orientations
	uint32_t* d_trans2 = nullptr;   // [orientations][columns]: their transmission values
	uint32_t n_orient_max = 1;      // most orientations any chunk has (2 for a single individual, 8 for a trio)
	uint8_t* d_sel = nullptr;
	uint32_t* d_guess = nullptr;
	uint32_t* d_bt_counters = nullptr;
	uint32_t n_spec = 0;
	size_t chunk_lds = 0;
	// slot runs (slots.h): the default forward path of a single individual
	SlotPlan splan;
	bool use_slots = false;
	int slot_l = 11;            // preferred number of local slots (lr + 6 .. lr + 9; pedigree runs: 6 - log2 T .. + 3, only when set explicitly)
	bool slot_l_set = false;
	int slot_lr = 2;            // reg slots: 4 cells per thread -> 8 waves per workgroup at 11 local slots (two waves per SIMD)
	int slot_lr_used = 2;       // ... of the plan in use (the shared-launches layout takes 3)
	std::vector<SlotBatchEntry> slot_entries;   // (pedigree runs: `pad` holds the run's index into splan.pextra)
	SlotBatchEntry* d_slot_entries = nullptr;
	uint64_t table_bytes = 0;   // pedigree slot runs: cost-form tables
	BtJob* d_btjobs = nullptr;
	uint32_t* d_job_scores = nullptr;
	bool shared_hint = false;   // option shared_launches: the table will be solved together with many others (enqueue_many)
	bool side_by_side = false;  // this solve's launches run beside other tables' on their own streams: X runs take the streamed variant (16 KB of LDS instead of ~90:
	                            // three tables on three streams, 3.1 M columns/s with the LDS lines -- one workgroup per CU, the streams take turns -- 5 M without)
	int max_lanes = 32;
	size_t next_super = 0;  // cursor of the resumable enqueue
	// Windowed solve (backtrace arena larger than what HBM can hold): the steps are cut into WINDOWS whose records fit the
	// arena one at a time.  Pass 1 runs the whole forward pass (records of all windows but the last are written and
	// dropped) and keeps the exchange column at every window boundary; then, newest window first, the window's steps
	// are run again from the kept column -- this time its records survive until its units have been walked.  Twice the
	// forward work, any table length.  Offsets into the arena are window-relative.
	struct Window {
		uint32_t step_lo = 0, step_hi = 0;   // positions in the job's step list
		uint32_t unit_off = 0, unit_count = 0;
	};
	uint64_t arena_limit = 0;   // option arena_limit_bytes (0: what hipMemGetInfo leaves)
	bool windowed = false;
	std::vector<Window> windows;
	uint8_t* d_checkpoints = nullptr;   // [windows - 1] exchange columns
	size_t checkpoint_bytes = 0;
	uint32_t* d_bt_state = nullptr;     // (x, transmission) the walk of a window hands to the next older one
	BtJob* d_window_jobs = nullptr;

	void launch_column_step(const Problem& p, const Step& step, const Lane& lane, const uint32_t* prev, uint32_t* cur, uint64_t& launches);
	void launch_run(const ResBatchEntry& e, uint32_t step_index, uint64_t& launches);
	void launch_slot_run(const SlotBatchEntry& e, uint64_t& launches);
	whamd_status_t begin_solve(const Problem& p, Solution& s, std::string& msg);
	whamd_status_t submit_singles(const Problem& p, const SuperStep& ss, uint64_t& launches, std::string& msg);
	whamd_status_t submit_tail(const Problem& p, std::string& msg, hipStream_t tail_stream = nullptr, bool backtrace_done = false, bool superreads_done = false);
	BtGroupEntry h_bt_entry{};          // what a batched backtrace launch reads for this table (kernels_backtrace.h backtrace_chunks_group), and its device copy
	BtGroupEntry* d_bt_entry = nullptr;
	// get_super_reads on the device (kernels_backtrace.h superreads_single): a table with one individual and trusted genotypes
	bool device_superreads = false;
	SuperreadArgs super_args{};
	size_t super_words = 0, super_off = 0;   // size of the result (u32 words) and where it lies in h_pinned
	bool own_stream_used = false;       // something has been submitted to `stream` since it was last synchronised (a member of a group solve never touches its own: its
	                                    // solve and tail are on the lead's stream, its uploads on an upload stream -- release() then has nothing to wait for there)
	bool upload_pending = false;        // the uploads went through a shared upload stream and no solve has been ordered behind `ev_upload` yet
	hipStream_t upload_stream = nullptr;
	bool timing_pending = false;        // wait() has collected a solve whose event timings nobody has read yet (read_timing)
	hipStream_t tail_stream = nullptr;  // where submit_tail put the tail of the solve in flight, and its place in the order of all tails of the process
	uint64_t tail_seq = 0;
	bool tail_elsewhere = false;        // the tail (backtrace, downloads) of the solve in flight went onto another table's stream: wait() waits for ev3, not for `stream`

	void release_lanes() {
		max_grid_x = 1;
		lanes.clear();
		jobs.clear();
		schedule.clear();
		entries.clear();
		slot_entries.clear();
		windows.clear();
	}

	void release() {
		if (tail_elsewhere && ev3) (void)hipEventSynchronize(ev3);   // (a solve whose tail ran on the group's lead stream: nothing of it may still be reading the buffers)
		tail_elsewhere = false;
		if (upload_pending && ev_upload) (void)hipEventSynchronize(ev_upload);   // (a table closed without a solve: its copies may still be on the upload stream)
		upload_pending = false;
		release_lanes();
		windowed = false;
		if (stream && own_stream_used && (!allocations.empty() || d_arena)) { (void)hipStreamSynchronize(stream); own_stream_used = false; }   // (hipFree used to wait for the table's last kernels)
		for (auto& a : allocations) devpool_give(device, a.first, a.second);
		allocations.clear();
		arena_give(device, d_arena, arena_bytes);
		d_arena = nullptr;
		arena_bytes = 0;
		pinned_give(h_pinned, h_pinned_bytes);
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
		release_device();
		delete impl_;
	}
}

void DeviceTable::release_device() {
	Impl& m = *impl_;
	(void)hipSetDevice(m.device);
	m.release();   // (synchronises the stream before anything is handed back)
	if (m.stream) {
		if (m.own_stream_used) (void)hipStreamSynchronize(m.stream);   // (13 ms for the 96 tables of a step when every table waited for a stream it had never used)
		m.own_stream_used = false;
		StreamSet ss;
		ss.stream = m.stream; ss.ev[0] = m.ev0; ss.ev[1] = m.ev1; ss.ev[2] = m.ev2; ss.ev[3] = m.ev3; ss.ev[4] = m.ev_group; ss.ev[5] = m.ev_upload; ss.device = m.device;
		streamset_give(ss);
	}
	m.stream = m.run_stream = nullptr;
	m.ev0 = m.ev1 = m.ev2 = m.ev3 = m.ev_group = m.ev_upload = nullptr;
	m.timing_pending = false;
}

int DeviceTable::device_count() {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

bool DeviceTable::device_pci_bus_id(int device, std::string& out) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return false;
	char text[64] = {0};
	if (hipDeviceGetPCIBusId(text, (int)sizeof text - 1, device) != hipSuccess) return false;
	out = text;
	for (char& ch : out) ch = (char)tolower((unsigned char)ch);   // (sysfs spells the id in lower case)
	return true;
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
	if (path != "auto" && path != "column" && path != "column_keys" && path != "resident" && path != "slots") return false;
	impl_->path = path;
	return true;
}

void DeviceTable::set_l_pref(int l) { impl_->l_pref = std::max(4, std::min(l, RES_LMAX)); }
void DeviceTable::set_lanes(int n) { impl_->max_lanes = n < 1 ? 1 : (n > 64 ? 64 : n); }

void DeviceTable::set_fold(bool v) { impl_->fold = v; }
void DeviceTable::set_slot_l(int l) { impl_->slot_l = std::max(2, std::min(l, 12)); impl_->slot_l_set = true; }
void DeviceTable::set_slot_lr(int lr) { impl_->slot_lr = lr >= 3 ? 3 : (lr <= 1 ? 1 : 2); }

void DeviceTable::set_arena_limit(uint64_t bytes) { impl_->arena_limit = bytes; }
void DeviceTable::set_shared_launches(bool v) { impl_->shared_hint = v; }
void DeviceTable::set_side_by_side(bool v) { impl_->side_by_side = v; }
void DeviceTable::set_symmetry(int level) { impl_->symmetry = level < 0 ? 0 : (level > 2 ? 2 : level); }

whamd_status_t DeviceTable::upload(Problem& p, int device, std::string& msg) {
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
	if (!m.stream) {
		StreamSet ss;
		if (!streamset_take(device, ss)) { msg = "could not create the table's stream and events"; return WHAMD_ERR_DEVICE; }
		m.stream = ss.stream; m.ev0 = ss.ev[0]; m.ev1 = ss.ev[1]; m.ev2 = ss.ev[2]; m.ev3 = ss.ev[3]; m.ev_group = ss.ev[4]; m.ev_upload = ss.ev[5];
	}
	m.release();
	const uint32_t n = p.n_cols;
	if (n == 0) return WHAMD_OK;
	// pedigrees beyond the templated kernels (three trios, more than six individuals): the generic per-column kernel, key path only
	m.wide = !select_kernels(p.T, p.n_ind, m.fused, m.keysfn);
	if (m.wide && (p.T > (uint32_t)MAX_T_WIDE || p.n_ind > (uint32_t)MAX_IND_WIDE)) {
		msg = "no device kernel for T=" + std::to_string(p.T) + ", individuals=" + std::to_string(p.n_ind);
		return WHAMD_ERR_UNSUPPORTED;
	}
	// ... and a column with more allele-assignment terms than the templated kernels stage in LDS (genotypes not trusted, seven and more
	// individuals): the generic kernel reads them from global memory
	for (uint32_t c = 0; c < p.n_cols && !m.wide; ++c) m.wide = p.term_end(c, p.T - 1) - p.term_begin(c, 0) > (uint64_t)COL_MAXTERMS;
	const bool force_keys = m.path == "column_keys" || m.wide;
	const bool want_resident = (m.path == "auto" || m.path == "resident") && !m.wide;
	const auto tu0 = std::chrono::steady_clock::now();
	size_t free_b = 0, total_b = 0;
	HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	if (free_b < total_b / 2) {   // a genotyping call of this process may be holding its column store (genotype.h): give it back first
		genotype_release_cache();
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	}
	if (free_b < total_b / 4) {   // tight: the arenas and buffers kept from closed tables go back as well
		dptable_release_arena_cache();
		devpool_release();
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	}
	free_b += arena_idle_bytes(device);   // (taken below, or freed before this table's own arena is allocated)
	// A wide single-individual table (coverage >= 18: 128 and more workgroups per launch) that will share its launches with many others
	// takes EIGHT cells per thread and twelve local slots: half the wavefronts per table and longer runs -- 24 coverage-20 tables
	// 7.7 M columns/s instead of 6.4 M; alone the same table is slower that way (1.87 M against 2.26 M), and narrow tables gain nothing.
	int slot_lr = m.slot_lr, slot_l = m.slot_l;
	if (m.shared_hint && p.T == 1 && p.max_k >= 18 && !m.slot_l_set && m.slot_lr == 2) { slot_lr = 3; slot_l = 12; }
	// Coverage 21 and 22 (512 / 1 024 workgroups of four cells per launch: the table fills the chip by itself, rounds of workgroups queue behind each
	// other) take the eight-cell layout ALONE as well: 14.0 against 17.9 us per launch at 21, 20.3 against 30.1 at 22 (scripts/gpu_wide_ab.py; the same
	// kernels with their operands streamed: 14.4 / 22.6).  At 23 the four-cell kernel with streamed operands wins (36.2 against 43.0 us): launch_slot_run.
	if (p.T == 1 && p.max_k >= 21 && p.max_k <= 22 && !m.slot_l_set && m.slot_lr == 2 && !debug_env("WHAMD_NO_WIDE_LAYOUT")) { slot_lr = 3; slot_l = 12; }
	m.slot_lr_used = slot_lr;
	m.use_slots = (m.path == "auto" || m.path == "slots") && !m.wide && plan_forward_slots(p, p.T > 1 ? (m.slot_l_set ? -m.slot_l : 0) : std::max(8, slot_l), m.symmetry, m.splan, slot_lr);
	// pedigree slot runs keep their cost-form tables in HBM (slots.h): at most a quarter of what is free, else the older paths
	if (m.use_slots && m.splan.ped && m.splan.table_words * 4ull > free_b / 4) m.use_slots = false;
	m.table_bytes = m.use_slots && m.splan.ped ? m.splan.table_words * 4ull : 0ull;
	if (p.lazy_terms) {
		// generic term lists where something will read them: the columns a pedigree slot plan leaves to the per-column kernels; every column on any other path
		std::vector<uint8_t> need(p.n_cols, 1);
		if (m.use_slots && m.splan.ped) for (uint32_t c = 0; c < p.n_cols; ++c) need[c] = m.splan.col_to_row[c] < 0;
		const whamd_status_t st = fill_lazy_terms(p, need, msg);
		if (st != WHAMD_OK) return st;
	}
	if (m.use_slots) {
		// the driver below walks plan.steps / plan.component_first_step; slot runs are steps of kind 2
		m.plan = ResidentPlan();
		m.plan.steps = m.splan.steps;
		m.plan.component_first_step = m.splan.component_first_step;
		m.plan.col_to_res.assign(p.n_cols, -1);
	} else {
		m.splan = SlotPlan();
		plan_forward(p, want_resident, m.l_pref, m.fold, m.plan, m.symmetry);
	}
	const auto tu1 = std::chrono::steady_clock::now();
	if (debug_env("WHAMD_DEBUG_PLAN")) {
		for (const Step& st : m.plan.steps) {
			if (st.kind == 0) { fprintf(stderr, "[plan] column %u k=%u b=%u f=%u\n", st.index, p.k[st.index], p.b[st.index], p.f[st.index]); continue; }
			if (st.kind == 2) {
				const SlotRun& r = m.splan.runs[st.index];
				fprintf(stderr, "[plan] slot run c0=%u ncols=%u g=%u L=%u half=%u ends=%u has_prev=%u in_identity=%u in_half=%u mirror_pos=%u in_occ=%x out_occ=%x mirror_out=%u\n",
				        r.c0, r.ncols, r.g, r.L, r.half, r.n_ends, r.has_prev, r.in_identity, r.in_half, r.in_mirror_pos, r.in_occ, r.out_occ, r.mirror_out);
				if (m.splan.ped) fprintf(stderr, "[plan]   pedigree run: T=%u forms per value=%u table words=%u record words per workgroup=%u\n", 1u << m.splan.pextra[st.index].tb,
				                         m.splan.pextra[st.index].nf, m.splan.pextra[st.index].s_off + r.ncols * 64u * pslot_ns(m.splan.pextra[st.index].nf), m.splan.pextra[st.index].rec_words);
				continue;
			}
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
	const bool timing = getenv("WHAMD_DEBUG_TIMING") != nullptr;
	auto lap_t = std::chrono::steady_clock::now();
	auto ulap = [&](const char* what) {
		if (!timing) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[whamd timing]   upload: %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - lap_t).count());
		lap_t = now;
	};
	// ---- descriptors
	m.cols.resize(n);
	std::vector<uint32_t> segs;
	RawVec<uint32_t> term_ptr32((size_t)n * (p.T + 1));
	static_assert(sizeof(CostTerm) == sizeof(DevTerm), "Problem::terms / fterms are uploaded as they are");
	const RawVec<CostTerm>& terms = p.terms;
	if (p.terms.size() >= 0xFFFFFFFFull || (uint64_t)p.col_ptr[n] * ni >= 0xFFFFFFFFull) {
		msg = "problem too large for 32-bit device offsets";
		return WHAMD_ERR_UNSUPPORTED;
	}
	uint64_t bt = 0, seg_bt = 0;
	size_t seg_cursor = 0, slot_cursor = 0;
	uint32_t max_f = 0, max_keys_f = 0;
	// what the arena may take: free HBM minus the descriptors (~1 KiB per column), exchange buffers, tables and slack
	const uint64_t reserve = (3ull << 30) + (uint64_t)n * 1024ull + m.table_bytes;
	const bool ped_slots = m.use_slots && m.splan.ped;
	auto slot_record_bytes = [&](size_t ri) -> uint64_t {   // record of one slot run: only launched workgroups write
		const SlotRun& run = m.splan.runs[ri];
		if (ped_slots) return (uint64_t)m.splan.pextra[ri].rec_words * 4ull << run.g;
		return (uint64_t)run.n_ends * run.threads * (1ull << (run.g - run.half));
	};
	uint64_t arena_cap = free_b > reserve ? free_b - reserve : 0;
	if (m.arena_limit) arena_cap = std::min<uint64_t>(arena_cap, m.arena_limit);
	std::vector<uint32_t> window_first_col;   // first column of every window after the first
	uint64_t bt_max = 0;
	auto open_unit = [&](uint32_t c, uint64_t bytes) -> bool {   // a backtrace unit of `bytes` starts at column c
		if (bytes + 16 > arena_cap) return false;
		if (bt + bytes + 16 > arena_cap) {
			bt_max = std::max(bt_max, bt);
			bt = 0;
			window_first_col.push_back(c);
		}
		return true;
	};
	auto unit_too_large = [&](uint32_t c) {
		msg = "the backtrace record of the unit at column " + std::to_string(c) + " alone does not fit in the arena (" + std::to_string(arena_cap >> 20) +
		      " MiB of " + std::to_string(free_b >> 20) + " MiB free HBM)";
		return WHAMD_ERR_UNSUPPORTED;
	};
	// what a column's descriptor holds by itself: a few host threads; the offsets that run through the table (segment lists,
	// backtrace records, windows) follow in column order
	parallel_ranges(n, host_threads(n, 16384), [&](uint64_t c0, uint64_t c1, uint32_t) {
		for (uint32_t c = (uint32_t)c0; c < (uint32_t)c1; ++c) {
			DevColumn d{};
			d.k = p.k[c];
			d.b = p.b[c];
			d.f = p.f[c];
			d.recomb = p.recomb[c];
			d.delta_off = (uint32_t)(p.col_ptr[c] * ni);
			d.term_off = (uint32_t)((size_t)c * (p.T + 1));
			for (uint32_t t = 0; t <= p.T; ++t) term_ptr32[(size_t)c * (p.T + 1) + t] = (uint32_t)p.term_ptr[(size_t)c * p.T + t];
			d.ebits = d.k - d.f;
			d.is_last = (c + 1 == n);
			d.eloop = std::min<uint32_t>(d.ebits, QMAX);
			d.nplanes = d.ebits + tbits;
			m.cols[c] = d;
		}
	});
	for (uint32_t c = 0; c < n; ++c) {
		DevColumn& d = m.cols[c];
		const bool in_slot_run = m.use_slots && m.splan.col_to_row[c] >= 0;   // (the run kernels read none of the per-column arrays)
		d.seg_off = (uint32_t)segs.size();
		const uint32_t kmask = d.k >= 32 ? 0xFFFFFFFFu : ((1u << d.k) - 1u);
		if (!in_slot_run) {
			append_segments(p.fwd_mask[c], segs, d.nseg_fwd);
			append_segments(kmask & ~p.fwd_mask[c], segs, d.nseg_end);
		}
		d.bt_off = bt;
		if (m.use_slots && m.splan.col_to_row[c] >= 0) {
			d.mode = 3;
			d.res_idx = (uint32_t)m.splan.col_to_row[c];
			if (slot_cursor < m.splan.runs.size() && m.splan.runs[slot_cursor].c0 == c) {  // first column of a slot run
				SlotRun& run = m.splan.runs[slot_cursor];
				if (!open_unit(c, slot_record_bytes(slot_cursor))) return unit_too_large(c);
				run.rec_lo = (uint32_t)bt;
				run.rec_hi = (uint32_t)(bt >> 32);
				seg_bt = bt;
				bt += slot_record_bytes(slot_cursor);
				max_f = std::max(max_f, run.L + run.g);   // entry / exit indices in physical order need 2^(L + g) entries
				++slot_cursor;
			}
			d.bt_off = seg_bt;
		} else if (m.plan.col_to_res[c] >= 0) {
			d.mode = 2;
			d.res_idx = (uint32_t)m.plan.col_to_res[c];
			if (seg_cursor < m.plan.segments.size() && m.plan.segments[seg_cursor].c0 == c) {  // first column of a run
				ResSegment& sgm = m.plan.segments[seg_cursor];
				if (!open_unit(c, (uint64_t)sgm.stage_words * (1ull << sgm.g) * 8ull)) return unit_too_large(c);
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
			const uint64_t bytes = d.mode == 0 ? (uint64_t)d.nplanes * p.T * (1ull << (d.f - 6)) * 8ull : (uint64_t)p.T * (1ull << d.f) * 4ull;
			if (!open_unit(c, bytes)) return unit_too_large(c);
			d.bt_off = bt;
			bt += bytes;
			if (d.mode != 0) max_keys_f = std::max(max_keys_f, d.f);
		}
		bt = (bt + 15ull) & ~15ull;
		max_f = std::max(max_f, d.f);
	}
	ulap("column descriptors (host)");
	bt = bt_max = std::max(bt_max, bt);
	m.bt_bytes = bt;
	m.windowed = !window_first_col.empty();
	m.checkpoint_bytes = (size_t)(1ull << max_f) * p.T * 4;
	const uint64_t need = bt + (2ull + window_first_col.size()) * (1ull << max_f) * p.T * 4ull + (1ull << max_keys_f) * p.T * 8ull;
	if (need + (1ull << 30) > free_b) {
		msg = "backtrace arena of " + std::to_string(need >> 20) + " MiB does not fit in free HBM (" + std::to_string(free_b >> 20) + " MiB)";
		return WHAMD_ERR_UNSUPPORTED;
	}
	// ---- allocate + upload
	// Everything this function sends or launches goes through one of the device's upload streams (upload_stream_of); begin_solve orders the solve behind ev_upload.
	// WHAMD_UPLOAD_ON_TABLE_STREAM=1 (debug library): the table's own stream, as before.
	hipStream_t us = debug_env("WHAMD_UPLOAD_ON_TABLE_STREAM") ? nullptr : upload_stream_of(device);
	if (!us) us = m.stream;
	m.upload_stream = us;
	if (us == m.stream) m.own_stream_used = true;
	StageSession stage(us);
	ulap("staging area taken");
	auto alloc = [&](void** dptr, size_t bytes) -> hipError_t {
		size_t got = 0;
		hipError_t e = devpool_take(device, std::max<size_t>(bytes, 16), dptr, &got);
		if (e == hipSuccess) m.allocations.emplace_back(*dptr, got);
		return e;
	};
	std::vector<int32_t> delta_fallback;
	const int32_t* delta_src = p.delta.data();
	size_t delta_count = (size_t)p.col_ptr[n] * p.n_ind;
	if (p.n_ind == 0) { delta_fallback.assign(std::max<size_t>(p.col_ptr[n], 1), 0); delta_src = delta_fallback.data(); delta_count = delta_fallback.size(); }
	void *d_delta, *d_term_ptr, *d_terms, *d_segs, *d_bt, *d_keys, *d_last_keys, *d_rcol, *d_rbt, *d_fterms = nullptr;
	// The superreads of a single-individual table with trusted genotypes are made on the device, behind the backtrace (the condition is finish_columns' first branch).
	m.device_superreads = n > 0 && p.n_ind == 1 && p.T == 1 && p.P == 2 && !p.distrust && p.h2p.size() >= 2 && p.h2p[0] == 0 && p.h2p[1] == 1 && p.genotype.size() >= n &&
	                      !debug_env("WHAMD_HOST_SUPERREADS");
	const size_t super_upload = m.device_superreads ? ((size_t)n + 1) * 8 + n + 1024 : 0;
	const size_t upload_bytes = m.cols.size() * sizeof(DevColumn) + delta_count * sizeof(int32_t) + term_ptr32.size() * 4 + (terms.size() + p.fterms.size()) * sizeof(DevTerm) + segs.size() * 4 +
	             m.plan.columns.size() * (sizeof(ResColumn) + sizeof(ResBacktrace) + sizeof(PedColumn)) + m.plan.ped_terms.size() * sizeof(PedTerm) +
	             (m.splan.rows.size() + SLOT_ROW_PAD) * sizeof(SlotRow) + m.splan.prows.size() * sizeof(PedSlotRow) + m.splan.bt_cols.size() * (sizeof(SlotBtCol) + 8) +
	             m.splan.runs.size() * (sizeof(SlotRun) + sizeof(PedSlotExtra) + sizeof(BtUnit) + sizeof(SlotBatchEntry) + 64) + m.plan.segments.size() * (sizeof(ResBatchEntry) + sizeof(BtUnit)) + super_upload + ((size_t)8 << 20);
	stage.expect(upload_bytes);
	// ONE device block and ONE staging image per table: every uploaded array is a piece of the block at the offset it has in the pinned area, and the pieces
	// leave as a few large copies.  (Per-array copies of ~1 MB ran at 25 GB/s -- 96 coverage-15 tables, 28 MB each, spent their creates waiting for the link --;
	// pieces of 32 MB and more reach 56 GB/s: scripts/micro/r6_h2d_rate.py.)
	char* d_slab = nullptr;
	size_t slab_cap = 0, slab_used = 0, slab_flushed = 0;
	if (upload_bytes <= STAGE_MAX && !debug_env("WHAMD_NO_UPLOAD_SLAB") && stage.begin_image(upload_bytes)) {
		void* ptr = nullptr;
		if (alloc(&ptr, upload_bytes) == hipSuccess) { d_slab = (char*)ptr; slab_cap = upload_bytes; }
		else { (void)hipGetLastError(); stage.image = false; }
	}
	ulap("staging image sized, device block taken");
	bool unstaged_copies = false;   // a copy whose source is pageable memory of this call: the create must wait for it
	const hipStream_t copy_stream = us;   // (all tables' images on ONE shared stream instead of sixteen at once was measured: 1 055 - 1 273 against 1 015 - 1 153 creates/s, noise)
	auto flush_slab = [&]() -> hipError_t {
		if (!d_slab || slab_used == slab_flushed) return hipSuccess;
		// (WHAMD_SKIP_SLAB_COPY=1, debug library, RESULTS INVALID: the image is built but does not travel -- what the creates cost without the link)
		const hipError_t e = debug_env("WHAMD_SKIP_SLAB_COPY") ? hipSuccess : hipMemcpyAsync(d_slab + slab_flushed, stage.base + slab_flushed, slab_used - slab_flushed, hipMemcpyHostToDevice, copy_stream);
		stage.pending = true;
		slab_flushed = slab_used;
		return e;
	};
	auto up = [&](void** dptr, const void* src, size_t bytes) -> hipError_t {
		const size_t padded = (bytes + 255) & ~(size_t)255;
		if (d_slab && slab_used + padded <= slab_cap) {
			*dptr = d_slab + slab_used;
			char* at = stage.base + slab_used;
			const char* from = (const char*)src;
			if (bytes >= ((size_t)4 << 20)) parallel_ranges(bytes, host_threads(bytes, (size_t)2 << 20), [&](uint64_t b0, uint64_t b1, uint32_t) { std::memcpy(at + b0, from + b0, b1 - b0); });
			else if (bytes) std::memcpy(at, from, bytes);
			slab_used += padded;
			stage.total += padded;
			return slab_used - slab_flushed >= ((size_t)32 << 20) ? flush_slab() : hipSuccess;
		}
		hipError_t e = alloc(dptr, bytes);
		if (e != hipSuccess) return e;
		if (bytes) e = stage.copy(*dptr, src, bytes);
		if (bytes && stage.image) unstaged_copies = true;   // (did not fit the image: copied straight from the caller's memory)
		return e;
	};
	// A single-individual table on slot runs: no kernel reads the per-column arrays (descriptor, deltas, term offsets, terms) of a column INSIDE a run -- the run
	// kernels and slot_tables work from the rows, the backtrace from its units -- so only the columns outside runs travel (the coverage ramp, the last column, what an
	// irregular layout leaves between runs): 96 of a column's 450 bytes, and concurrent creates are bound by bytes through the link.  The arrays keep
	// their size and indexing on the device; the pieces that are not sent are never read.
	std::vector<std::pair<uint32_t, uint32_t>> sent;   // [first, last) column ranges that are uploaded
	const bool sparse_columns = m.use_slots && !ped_slots && p.T == 1 && !m.windowed && !debug_env("WHAMD_DENSE_COLUMN_UPLOAD");
	if (sparse_columns) {
		for (uint32_t c = 0; c < n;) {
			if (m.splan.col_to_row[c] >= 0) { ++c; continue; }
			uint32_t e = c + 1;
			while (e < n && m.splan.col_to_row[e] < 0) ++e;
			sent.emplace_back(c, e);
			c = e;
		}
	}
	// reserves [count x elem] bytes like `up`, sends only the element ranges of `pieces` (element index = f(column))
	auto up_pieces = [&](void** dptr, const void* src, size_t bytes, const std::vector<std::pair<size_t, size_t>>& pieces) -> hipError_t {
		const size_t padded = (bytes + 255) & ~(size_t)255;
		const bool in_slab = d_slab && slab_used + padded <= slab_cap;
		if (in_slab) {
			hipError_t e = flush_slab();   // what is staged so far leaves as it is; this array's pieces go out on their own
			if (e != hipSuccess) return e;
			*dptr = d_slab + slab_used;
			slab_used += padded;
			slab_flushed = slab_used;      // (nothing of this array is in the staging image)
		} else {
			hipError_t e = alloc(dptr, bytes);
			if (e != hipSuccess) return e;
		}
		const size_t image_at = slab_used - padded;   // (in_slab: where the array lies in the block AND in the staging image)
		for (const auto& pc : pieces) {
			if (pc.second <= pc.first) continue;
			const char* from = (const char*)src + pc.first;
			if (in_slab) {   // through the pinned image, like everything else: the copy's source outlives the create
				std::memcpy(stage.base + image_at + pc.first, from, pc.second - pc.first);
				from = stage.base + image_at + pc.first;
				stage.pending = true;
			} else {
				unstaged_copies = true;   // (straight from the caller's pageable memory: upload() ends with a host wait)
			}
			hipError_t e = hipMemcpyAsync((char*)*dptr + pc.first, from, pc.second - pc.first, hipMemcpyHostToDevice, us);
			if (e != hipSuccess) return e;
		}
		return hipSuccess;
	};
	if (sparse_columns) {
		std::vector<std::pair<size_t, size_t>> pc_cols, pc_delta, pc_tptr, pc_terms;
		for (const auto& r : sent) {
			pc_cols.emplace_back((size_t)r.first * sizeof(DevColumn), (size_t)r.second * sizeof(DevColumn));
			pc_delta.emplace_back((size_t)p.col_ptr[r.first] * p.n_ind * sizeof(int32_t), (size_t)p.col_ptr[r.second] * p.n_ind * sizeof(int32_t));
			pc_tptr.emplace_back((size_t)r.first * (p.T + 1) * sizeof(uint32_t), (size_t)r.second * (p.T + 1) * sizeof(uint32_t));
			pc_terms.emplace_back((size_t)p.term_ptr[(size_t)r.first * p.T] * sizeof(DevTerm), (size_t)p.term_ptr[(size_t)r.second * p.T] * sizeof(DevTerm));
		}
		HIP_TRY(up_pieces((void**)&m.d_cols, m.cols.data(), m.cols.size() * sizeof(DevColumn), pc_cols));
		// (the deltas travel whole when the device makes the superreads: superreads_single reads every column's)
		if (m.device_superreads) HIP_TRY(up(&d_delta, delta_src, delta_count * sizeof(int32_t)));
		else HIP_TRY(up_pieces(&d_delta, delta_src, delta_count * sizeof(int32_t), pc_delta));
		HIP_TRY(up_pieces(&d_term_ptr, term_ptr32.data(), term_ptr32.size() * sizeof(uint32_t), pc_tptr));
		HIP_TRY(up_pieces(&d_terms, terms.data(), terms.size() * sizeof(DevTerm), pc_terms));
	} else {
	HIP_TRY(up((void**)&m.d_cols, m.cols.data(), m.cols.size() * sizeof(DevColumn)));
	HIP_TRY(up(&d_delta, delta_src, delta_count * sizeof(int32_t)));
	HIP_TRY(up(&d_term_ptr, term_ptr32.data(), term_ptr32.size() * sizeof(uint32_t)));
	HIP_TRY(up(&d_terms, terms.data(), terms.size() * sizeof(DevTerm)));
	}
	if (m.device_superreads) {
		void *d_cp = nullptr, *d_geno = nullptr;
		HIP_TRY(up(&d_cp, p.col_ptr.data(), ((size_t)n + 1) * sizeof(uint64_t)));
		HIP_TRY(up(&d_geno, p.genotype.data(), (size_t)n));
		m.super_args = SuperreadArgs{};
		m.super_args.delta = (const int32_t*)d_delta;
		m.super_args.col_ptr = (const unsigned long long*)d_cp;
		m.super_args.genotype = (const uint8_t*)d_geno;
		m.super_args.n_cols = n;
		m.super_words = (size_t)n + ((size_t)2 * n + 3) / 4;
	}
	if (!p.fterms.empty()) HIP_TRY(up(&d_fterms, p.fterms.data(), p.fterms.size() * sizeof(DevTerm)));   // factorised lines (pedslot_tables, PSLOT_FACT)
	HIP_TRY(up(&d_segs, segs.data(), segs.size() * sizeof(uint32_t)));
	HIP_TRY(up(&d_rcol, m.plan.columns.data(), m.plan.columns.size() * sizeof(ResColumn)));
	HIP_TRY(up(&d_rbt, m.plan.backtrace.data(), m.plan.backtrace.size() * sizeof(ResBacktrace)));
	void *d_pcol = nullptr, *d_pterm = nullptr;
	m.plan.ped_columns.resize(m.plan.ped_columns.empty() ? 0 : m.plan.columns.size());
	HIP_TRY(up(&d_pcol, m.plan.ped_columns.data(), m.plan.ped_columns.size() * sizeof(PedColumn)));
	HIP_TRY(up(&d_pterm, m.plan.ped_terms.data(), m.plan.ped_terms.size() * sizeof(PedTerm)));
	m.dp.ped_cols = (const PedColumn*)d_pcol;
	m.dp.ped_terms = (const PedTerm*)d_pterm;
	ulap("column arrays: allocations + copies");
	// slot runs: per-column descriptors and the backtrace blobs ([ncols] SlotBtCol + ending slots per run)
	RawVec<uint32_t> slot_blob;
	std::vector<uint32_t> slot_blob_off(m.splan.runs.size(), 0), slot_blob_words(m.splan.runs.size(), 0);
	{
		// offsets first (one pass over the runs), then every run copies its own piece (8 MB for configs[2]: 0.9 ms when it was one growing vector)
		size_t words = 0;
		for (size_t ri = 0; ri < m.splan.runs.size(); ++ri) {
			const SlotRun& run = m.splan.runs[ri];
			slot_blob_off[ri] = (uint32_t)words;
			slot_blob_words[ri] = (uint32_t)((size_t)run.ncols * (sizeof(SlotBtCol) / 4) + (run.n_ends + 3) / 4 + 1);
			words += slot_blob_words[ri];
		}
		slot_blob.resize(words);
		parallel_ranges(m.splan.runs.size(), host_threads(m.splan.runs.size(), 512), [&](uint64_t r0, uint64_t r1, uint32_t) {
			for (size_t ri = r0; ri < r1; ++ri) {
				const SlotRun& run = m.splan.runs[ri];
				uint32_t* dst = slot_blob.data() + slot_blob_off[ri];
				const size_t col_words = (size_t)run.ncols * (sizeof(SlotBtCol) / 4);
				std::memcpy(dst, m.splan.bt_cols.data() + run.row_off, col_words * 4);
				const size_t end_words = (run.n_ends + 3) / 4 + 1;
				std::memset(dst + col_words, 0, end_words * 4);
				if (run.n_ends) std::memcpy(dst + col_words, m.splan.end_slots.data() + m.splan.end_off[ri], run.n_ends);
			}
		});
	}
	ulap("slot backtrace blobs (host)");
	void *d_srows = nullptr, *d_sblob = nullptr;
	if (!m.splan.rows.empty()) m.splan.rows.resize(m.splan.rows.size() + SLOT_ROW_PAD);   // the kernel's scalar-cache warm-up touches a fixed number of rows
	HIP_TRY(up(&d_srows, m.splan.rows.data(), m.splan.rows.size() * sizeof(SlotRow)));
	ulap("slot rows: allocation + copy");
	HIP_TRY(up(&d_sblob, slot_blob.data(), slot_blob.size() * sizeof(uint32_t)));
	void* d_sctrl = nullptr;
	HIP_TRY(up(&d_sctrl, m.splan.ctrl.data(), m.splan.ctrl.size() * sizeof(uint32_t)));
	m.dp.slot_ctrl = (const uint32_t*)d_sctrl;
	// single-individual slot runs: the prologue's tables (SlotRun::tab_g / tab_w / tab_sl), built on the device below
	void *d_sruns = nullptr, *d_stab = nullptr;
	uint64_t slot_tab_words = 0;
	if (m.use_slots && !ped_slots) {
		for (SlotRun& run : m.splan.runs) {
			auto pad4 = [](uint64_t v) { return (v + 3ull) & ~3ull; };
			// X runs (kernels_slots.h, slot_runx_body): a Y-form run with four cells per thread whose columns and ending reads fit the kernel's registers; the
			// rows of its tables are padded with zero columns to a pair of trips
			const bool xrun = (run.yflags & 1u) && (run.lr == 2u || (run.lr == 3u && !debug_env("WHAMD_NO_XRUN8"))) && run.ncols <= (uint32_t)SLOT_XCOLS &&
			                  run.n_ends <= (uint32_t)SLOT_XENDS && !debug_env("WHAMD_NO_XRUN");
			const uint64_t ncp = xrun ? ((run.ncols + 7u) & ~7u) : run.ncols;
			run.tab_g = (uint32_t)slot_tab_words;
			slot_tab_words += pad4(ncp << (run.g - run.half));
			run.tab_w = (uint32_t)slot_tab_words;
			slot_tab_words += pad4(ncp * (run.threads >> 6));
			run.tab_sl = (uint32_t)slot_tab_words;
			slot_tab_words += ncp * 64u;
			if (xrun) {
				run.yflags |= 8u;
				slot_tab_words = (slot_tab_words + 15u) & ~(uint64_t)15;   // (a trip's sixteen Kr words are ONE 64-byte line of the scalar cache)
				run.tab_kr = (uint32_t)slot_tab_words;
				slot_tab_words += pad4((((uint64_t)run.ncols + SLOT_XPAD) << run.lr) + run.ncols + SLOT_XPAD);   // Kr words, then one control word per column
				run.tab_par = (uint32_t)slot_tab_words;
				slot_tab_words += pad4((uint64_t)run.threads + (1ull << (run.g - run.half)));
			}
		}
		slot_tab_words += (uint64_t)SLOT_XCOLS * 64u;   // (an X run requests the lane parts of SLOT_XCOLS columns whatever its length)
		if (getenv("WHAMD_DEBUG_TIMING")) {
			size_t nx = 0, not_y = 0, not_lr = 0, long_run = 0, many_ends = 0;
			for (const SlotRun& run : m.splan.runs) {
				nx += (run.yflags & 8u) != 0;
				not_y += !(run.yflags & 1u); not_lr += run.lr != 2u && run.lr != 3u; long_run += run.ncols > (uint32_t)SLOT_XCOLS; many_ends += run.n_ends > (uint32_t)SLOT_XENDS;
			}
			fprintf(stderr, "[whamd timing]   X runs: %zu of %zu (not Y form %zu, cells per thread %zu, more than %d columns %zu, more than %d ending reads %zu)\n", nx, m.splan.runs.size(), not_y,
			        not_lr, SLOT_XCOLS, long_run, SLOT_XENDS, many_ends);
		}
		if (slot_tab_words >= 0xFFFFFFFFull) { msg = "slot-run tables exceed 32-bit offsets"; return WHAMD_ERR_UNSUPPORTED; }
		HIP_TRY(up(&d_sruns, m.splan.runs.data(), m.splan.runs.size() * sizeof(SlotRun)));
		ulap("slot blobs, control words, runs: allocations + copies");
		HIP_TRY(alloc(&d_stab, slot_tab_words * 4));
		ulap("slot tables: allocation");
	}
	m.dp.slot_tab = (const uint32_t*)d_stab;
	void *d_prows = nullptr, *d_pruns = nullptr, *d_pextra = nullptr, *d_ptab = nullptr;
	if (ped_slots) {
		HIP_TRY(up(&d_prows, m.splan.prows.data(), m.splan.prows.size() * sizeof(PedSlotRow)));
		HIP_TRY(up(&d_pruns, m.splan.runs.data(), m.splan.runs.size() * sizeof(SlotRun)));
		HIP_TRY(up(&d_pextra, m.splan.pextra.data(), m.splan.pextra.size() * sizeof(PedSlotExtra)));
		HIP_TRY(alloc(&d_ptab, m.table_bytes));
	}
	m.dp.pslot_rows = (const PedSlotRow*)d_prows;
	m.dp.pslot_tab = (const uint32_t*)d_ptab;
	m.dp.slot_rows = (const SlotRow*)d_srows;
	m.dp.slot_blob = (const uint32_t*)d_sblob;
	ulap("slot rows / blobs / control words: allocations + copies");
	// ---- jobs (see Impl::Job): connected components made of runs only get their own job
	m.release_lanes();
	{
		Impl::Job final_job;
		final_job.final = true;
		std::vector<Impl::Job> component_jobs;
		const std::vector<uint32_t>& first = m.plan.component_first_step;
		// Components as jobs of their own run side by side on lanes -- and walk back one after the other through the SEQUENTIAL backtrace (the chunked one takes a
		// single job): 16 ms for an irregular coverage-20 table of 200 000 columns whose forward pass is 57 ms, nearly all of it one giant component (a Poisson layout
		// leaves a gap every few ten thousand columns).  A table whose largest component holds four fifths of its steps or more stays ONE job: the lanes would gain
		// less than the walk loses ((1 - s) x forward against s x 16 ms).  WHAMD_SPLIT_COMPONENTS=1 (debug library): always split, as before.
		size_t largest = 0;
		for (size_t k = 0; k < first.size(); ++k) largest = std::max<size_t>(largest, (k + 1 < first.size() ? first[k + 1] : m.plan.steps.size()) - first[k]);
		const bool worth_splitting = largest * 5 < m.plan.steps.size() * 4 || debug_env("WHAMD_SPLIT_COMPONENTS");
		const bool split = m.max_lanes > 1 && first.size() > 1 && worth_splitting && !debug_env("WHAMD_DEBUG_STAMPS") && !m.windowed;
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
			} else if (st.kind == 2) {
				const SlotRun& run = m.splan.runs[st.index];
				SlotBtUnit su{};
				su.kind = ped_slots ? 3 : 2; su.c0 = run.c0; su.ncols = run.ncols; su.blob_off = slot_blob_off[st.index];
				su.g = run.g; su.L = run.L; su.n_ends = run.n_ends; su.threads = run.threads;
				su.bt_lo = run.rec_lo; su.bt_hi = run.rec_hi; su.half = run.half; su.blob_words = slot_blob_words[st.index];
				su.f_exit = m.splan.f_exit[st.index];
				for (uint32_t j = 0; j < su.f_exit && j < 32; ++j) su.exit_pos[j] = (uint8_t)slot_pos(run.out_pos, m.splan.exit_slot[st.index][j]);
				su.lr = run.lr;
				if (ped_slots) { su.n_ends = m.splan.pextra[st.index].rec_words; su.lr = m.splan.pextra[st.index].tb; }   // kind 3: words of one workgroup's record, log2 T
				for (uint32_t j = 0; j < su.f_exit && j < 32; ++j) su.exit_slot[j] = m.splan.exit_slot[st.index][j];
				static_assert(sizeof(SlotBtUnit) == sizeof(BtUnit), "unit headers share one array");
				std::memcpy(&u, &su, sizeof u);
			} else {
				const ResSegment& sgm = m.plan.segments[st.index];
				u.c0 = sgm.c0; u.ncols = sgm.ncols; u.col_off = sgm.col_off; u.g = sgm.g; u.Lf_last = sgm.Lf_last;
				u.stage_words = sgm.stage_words; u.n_wext = sgm.n_wext; u.bt_lo = sgm.bt_lo; u.bt_hi = sgm.bt_hi;
				u.n_lext = sgm.n_lext;
				u.pad0 = (uint32_t)sgm.bt_active | ((uint32_t)sgm.bt_simple << 16) | (sgm.half << 20);
				std::copy(sgm.wext, sgm.wext + RES_IOSEG, u.wext);
				std::copy(sgm.lext, sgm.lext + RES_BT_LRUNS, u.lext);
			}
			m.units.push_back(u);
		}
		job.unit_count = (uint32_t)m.units.size() - job.unit_off;
		}
	}
	// ---- windows (see Impl::Window): the single job's steps cut where the arena offsets start over
	if (m.windowed) {
		const std::vector<uint32_t>& steps = m.jobs[0].steps;
		const uint32_t n_steps = (uint32_t)steps.size();
		auto first_col = [&](uint32_t pos) {
			const Step& st = m.plan.steps[steps[pos]];
			return st.kind == 0 ? st.index : (st.kind == 2 ? m.splan.runs[st.index].c0 : m.plan.segments[st.index].c0);
		};
		Impl::Window w;
		size_t next = 0;
		for (uint32_t pos = 0; pos < n_steps; ++pos) {
			if (next < window_first_col.size() && first_col(pos) == window_first_col[next]) {
				w.step_hi = pos;
				m.windows.push_back(w);
				w = Impl::Window();
				w.step_lo = pos;
				++next;
			}
		}
		w.step_hi = n_steps;
		m.windows.push_back(w);
		if (next != window_first_col.size()) { msg = "internal error: a window does not start at a step"; return WHAMD_ERR_DEVICE; }
		std::vector<BtJob> wjobs;
		for (size_t wi = 0; wi < m.windows.size(); ++wi) {   // units are the steps in reverse order
			Impl::Window& win = m.windows[wi];
			win.unit_off = n_steps - win.step_hi;
			win.unit_count = win.step_hi - win.step_lo;
			wjobs.push_back(BtJob{win.unit_off, win.unit_count, wi + 1 == m.windows.size() ? 1u : 2u, 0u});
		}
		void* d_wjobs = nullptr;
		HIP_TRY(up(&d_wjobs, wjobs.data(), wjobs.size() * sizeof(BtJob)));
		m.d_window_jobs = (BtJob*)d_wjobs;
		HIP_TRY(alloc((void**)&m.d_checkpoints, (m.windows.size() - 1) * m.checkpoint_bytes));
		HIP_TRY(alloc((void**)&m.d_bt_state, 16));
		if (getenv("WHAMD_DEBUG_TIMING")) fprintf(stderr, "[whamd timing] windowed solve: %zu windows, arena %.2f GB\n", m.windows.size(), (double)bt / 1e9);
	}
	// ---- everything a solve hands back lies in ONE device block, laid out like the pinned buffer it is downloaded into: [n] path index, [n] path transmission, the
	// final job's score with the other jobs' behind it (d_job_scores[0] IS d_score: job 0 is the final one and has no entry of its own), four words of backtrace
	// counters, the superreads.  Five copies per table -- 4.6 us each as blit kernels, one after the other on a group's stream: 2.6 ms behind a 96-table solve -- are one.
	m.super_off = 2 * (size_t)n + 4 + m.jobs.size();
	{
		void* d_res = nullptr;
		HIP_TRY(alloc(&d_res, (m.super_off + (m.device_superreads ? m.super_words : 0)) * sizeof(uint32_t) + 16));
		uint32_t* res = (uint32_t*)d_res;
		m.d_path_index = res;
		m.d_path_trans = res + n;
		m.d_score = res + 2 * (size_t)n;
		m.d_job_scores = res + 2 * (size_t)n;
		m.d_bt_counters = res + 2 * (size_t)n + m.jobs.size();
		if (m.device_superreads) {
			m.super_args.out = res + m.super_off;
			m.super_args.path_index = m.d_path_index;
		}
	}
	// ---- chunks of the speculative backtrace: a new chunk starts at every BT_CHUNK_RUNS-th slot run (single job only)
	m.use_chunks = false;
	m.chunks.clear();
	m.n_spec = 0;
	for (SlotRun& run : m.splan.runs) run.spec_id = 0;
	const bool trio_runs = !m.use_slots && !m.plan.ped_columns.empty();   // LDS-resident trio runs (kernels_trio.h)
	if (trio_runs) for (ResSegment& sgm : m.plan.segments) sgm.in_mirror_bit = 0;   // (trio runs have no mirror: the field carries the spec id)
	if ((m.use_slots || trio_runs) && m.jobs.size() == 1 && !m.windowed && !getenv("WHAMD_BT_SEQUENTIAL") && m.units.size() > 2u * BT_CHUNK_RUNS) {
		m.use_chunks = true;
		// orientation generators (BtChunk): per individual the transmission bits its relabelling flips -- a founder those of the
		// trios it is a parent of, a child both of its own trio (exact where genotypes are heterozygous); individuals that are
		// both, or pedigrees with more than BT_GENERATORS individuals, get no generator (still exact, more walked twice)
		std::vector<int> generator_of(std::max<uint32_t>(p.n_ind, 1), -1);
		std::vector<uint32_t> generator_tflip;
		if (p.n_ind >= 2 && p.n_ind <= BT_GENERATORS) {
			for (uint32_t s = 0; s < p.n_ind; ++s) {
				uint32_t as_parent = 0, as_child = 0;
				for (uint32_t t3 = 0; t3 < p.n_triples; ++t3) {
					if (p.triples[t3][0] == s) as_parent |= 1u << (2 * t3);
					if (p.triples[t3][1] == s) as_parent |= 1u << (2 * t3 + 1);
					if (p.triples[t3][2] == s) as_child |= 3u << (2 * t3);
				}
				if (as_parent && as_child) continue;
				if (!as_parent && !as_child) continue;   // unrelated individual of a multi-sample table
				generator_of[s] = (int)generator_tflip.size();
				generator_tflip.push_back(as_parent | as_child);
			}
		}
		auto orientations = [&](BtChunk& ch, uint32_t unit) {
			// the chunk starts from the projection column of the LAST column of unit `unit`'s step: bit j = j-th forwarded read
			const BtUnit& bu = m.units[unit];
			const uint32_t c_last = bu.c0 + bu.ncols - 1;
			ch.n_orient = 1;
			for (uint32_t q = 0; q < BT_GENERATORS; ++q) ch.flip[q] = 0;
			if (p.n_triples == 0 && p.n_ind <= 1) {
				const uint32_t fb = p.f[c_last];
				ch.flip[0] = fb >= BT_STATE_TSHIFT ? BT_STATE_XMASK : ((1u << fb) - 1u);
				ch.n_orient = 2;
				return;
			}
			if (generator_tflip.empty()) return;
			const ColumnEntry* col = p.col_begin(c_last);
			uint32_t bit = 0;
			for (uint32_t j = 0; j < p.k[c_last]; ++j) {
				if (!((p.fwd_mask[c_last] >> j) & 1u)) continue;
				const int gq = generator_of[col[j].sample];
				if (gq >= 0) ch.flip[gq] |= 1u << bit;
				++bit;
			}
			for (size_t q = 0; q < generator_tflip.size(); ++q) ch.flip[q] |= generator_tflip[q] << BT_STATE_TSHIFT;
			ch.n_orient = 1u << generator_tflip.size();
		};
		BtChunk cur{};
		cur.n_orient = 1;   // the newest chunk starts from the table's optimum
		uint32_t runs_in_chunk = 0;
		for (uint32_t u = 0; u < m.units.size(); ++u) {
			const bool is_run = m.units[u].kind == 2 || m.units[u].kind == 3 || (trio_runs && m.units[u].kind == 1);
			if (is_run && runs_in_chunk >= (uint32_t)BT_CHUNK_RUNS && u > 0) {
				m.chunks.push_back(cur);
				cur = BtChunk{};
				cur.unit_off = u;
				cur.spec_id = ++m.n_spec;
				orientations(cur, u);
				runs_in_chunk = 0;
				// the run of unit u leaves the seed: units are the job's steps in reverse order
				const Step& st = m.plan.steps[m.jobs[0].steps[m.jobs[0].steps.size() - 1 - u]];
				if (trio_runs) m.plan.segments[st.index].in_mirror_bit = m.n_spec;
				else m.splan.runs[st.index].spec_id = m.n_spec;
			}
			runs_in_chunk += is_run;
			++cur.unit_count;
		}
		m.chunks.push_back(cur);
		void *d_chunks = nullptr;
		HIP_TRY(up(&d_chunks, m.chunks.data(), m.chunks.size() * sizeof(BtChunk)));
		m.d_chunks = (BtChunk*)d_chunks;
		m.n_orient_max = 1;
		for (const BtChunk& ch : m.chunks) m.n_orient_max = std::max(m.n_orient_max, ch.n_orient);
		HIP_TRY(alloc((void**)&m.d_unit_x, (size_t)m.n_orient_max * m.units.size() * 4));
		HIP_TRY(alloc((void**)&m.d_path2, (size_t)m.n_orient_max * n * 4));
		HIP_TRY(alloc((void**)&m.d_trans2, (size_t)m.n_orient_max * n * 4));
		HIP_TRY(alloc((void**)&m.d_sel, m.units.size() + 16));
		HIP_TRY(alloc((void**)&m.d_guess, m.chunks.size() * 4));
		uint32_t stride = 64;
		for (const SlotRun& run : m.splan.runs) if (run.spec_id) stride = std::max(stride, (run.threads >> 6) << (run.g - run.half));
		if (trio_runs) for (const ResSegment& sgm : m.plan.segments) if (sgm.in_mirror_bit) stride = std::max(stride, (sgm.threads >> 6) << sgm.g);
		m.dp.spec_stride = stride;
		void* d_spec = nullptr;
		HIP_TRY(alloc(&d_spec, ((size_t)m.n_spec + 1) * stride * 8));
		m.dp.spec_keys = (unsigned long long*)d_spec;
	} else {
		m.dp.spec_keys = nullptr;
	}
	HIP_TRY(up((void**)&m.d_units, m.units.data(), m.units.size() * sizeof(BtUnit)));
	{
		uint32_t max_stage = 0;
		for (const ResSegment& sgm : m.plan.segments) max_stage = std::max(max_stage, sgm.stage_words);
		for (size_t ri = 0; ri < m.splan.runs.size(); ++ri)
			max_stage = std::max(max_stage, ped_slots ? m.splan.pextra[ri].rec_words / 2u : (m.splan.runs[ri].n_ends * m.splan.runs[ri].threads + 7) / 8);
		m.bt_lds = (size_t)2 * RES_MAXCOLS * 128 + 512 + 16 + (size_t)BT_CELLS * 4 + (size_t)RES_MAXCOLS * 4 + (size_t)max_stage * 8 + 16;
		m.chunk_lds = (size_t)(32 + 4 + BT_CELLS + BT_CHUNK_BLOB) * 4 + (size_t)max_stage * 8 + 16;
	}
	void* d_rtab = nullptr;
	const bool ped_plan = !m.plan.ped_columns.empty();
	HIP_TRY(alloc(&d_rtab, m.plan.columns.size() * (ped_plan ? PED_TABLE : RES_TABLE) * sizeof(int32_t)));
	m.dp.res_tables = ped_plan ? nullptr : (int32_t*)d_rtab;
	m.dp.ped_tables = ped_plan ? (int32_t*)d_rtab : nullptr;
	ulap("jobs, backtrace units, schedule");
	const auto tu2 = std::chrono::steady_clock::now();
	{
		size_t got = 0;
		d_bt = arena_take(device, bt, got);
		if (!d_bt) {
			HIP_TRY(hipMalloc(&d_bt, std::max<size_t>(bt, 16)));
			got = std::max<size_t>(bt, 16);
		}
		m.d_arena = d_bt;
		m.arena_bytes = got;
	}
	const auto tu3 = std::chrono::steady_clock::now();
	m.key_entries = (size_t)(1ull << max_keys_f) * p.T;
	HIP_TRY(alloc(&d_keys, m.key_entries * 8));
	HIP_TRY(alloc(&d_last_keys, (size_t)MAX_T_WIDE * 8));
	HIP_TRY(alloc((void**)&m.d_pr[0], (size_t)(1ull << max_f) * p.T * 4));
	HIP_TRY(alloc((void**)&m.d_pr[1], (size_t)(1ull << max_f) * p.T * 4));
	HIP_TRY(pinned_take((m.super_off + (m.device_superreads ? m.super_words : 0)) * sizeof(uint32_t), (void**)&m.h_pinned, &m.h_pinned_bytes));
	{   // lanes: longest job first to the least loaded lane; lane 0 always runs the final job
		// at most 1 GiB of private exchange buffers (coverage 23: 64 MiB per lane)
		const size_t lane_bytes = 2 * ((size_t)(1ull << max_f) * p.T * 4);
		const size_t by_memory = std::max<size_t>(1, ((size_t)1 << 30) / std::max<size_t>(lane_bytes, 1));
		const size_t n_lanes = std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)m.max_lanes, by_memory), m.jobs.size()));
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
		m.lanes[0].d_pr[0] = m.d_pr[0];
		m.lanes[0].d_pr[1] = m.d_pr[1];
		m.lanes[0].d_keys = (unsigned long long*)d_keys;
		for (size_t l = 1; l < n_lanes; ++l) {
			HIP_TRY(alloc((void**)&m.lanes[l].d_pr[0], (size_t)(1ull << max_f) * p.T * 4));
			HIP_TRY(alloc((void**)&m.lanes[l].d_pr[1], (size_t)(1ull << max_f) * p.T * 4));
			HIP_TRY(alloc((void**)&m.lanes[l].d_keys, m.key_entries * 8));
		}
	}
	{   // schedule: the lanes advance in lockstep; super-step t holds step t of every lane that still has one
		struct Cursor { size_t job_i = 0, step_i = 0; uint32_t flip = 0; };
		std::vector<Cursor> cur(m.lanes.size());
		for (;;) {
			Impl::SuperStep ss;
			ss.entry_off = (uint32_t)(m.use_slots ? m.slot_entries.size() : m.entries.size());
			for (size_t li = 0; li < m.lanes.size(); ++li) {
				Impl::Lane& lane = m.lanes[li];
				Cursor& c = cur[li];
				if (c.job_i == lane.jobs.size()) continue;
				const uint32_t job_id = lane.jobs[c.job_i];
				const Impl::Job& job = m.jobs[job_id];
				const uint32_t si = job.steps[c.step_i];
				const bool first = c.step_i == 0, last = c.step_i + 1 == job.steps.size();
				const Step& step = m.plan.steps[si];
				if (step.kind == 2) {
					SlotBatchEntry e{};
					e.run = m.splan.runs[step.index];
					if (first) e.run.has_prev = 0;  // a job starts from cost 0
					e.prev = lane.d_pr[c.flip];
					e.cur = lane.d_pr[c.flip ^ 1];
					e.score_out = (last && !job.final) ? m.d_job_scores + job_id : nullptr;
					ss.io[0] = lane.d_pr[c.flip]; ss.io[1] = lane.d_pr[c.flip ^ 1];
					if (m.splan.ped) {
						const PedSlotExtra& pex = m.splan.pextra[step.index];
						ss.lds = std::max<size_t>(ss.lds, pedslot_lds_bytes(e.run.threads, e.run.ncols, pex));
					} else
					ss.lds = std::max<size_t>(ss.lds, slot_run_lds_bytes(e.run.threads, e.run.lr, e.run.ncols));
					e.pad = step.index;
					// the owning table's arrays: a group launch (slot_group / pedslot_group) serves runs of several tables
					if (ped_slots) e.ex = m.splan.pextra[step.index];
					e.tab = ped_slots ? (const uint32_t*)d_ptab : (const uint32_t*)d_stab;
					e.rows = ped_slots ? (const void*)d_prows : (const void*)d_srows;
					e.ctrl = (const uint32_t*)d_sctrl;
					e.bt = (uint8_t*)d_bt;
					e.spec_keys = m.dp.spec_keys;
					e.spec_stride = m.dp.spec_stride;
					if (const char* skip = debug_env("WHAMD_SLOT_SKIP")) e.pad2 = (uint32_t)atoi(skip);   // (timing experiments in a group launch)
					if (debug_env("WHAMD_NO_WARM")) e.pad2 |= 0x10000u;
					ss.grid_x = std::max(ss.grid_x, 1u << (e.run.g - e.run.half));
					ss.threads = std::max(ss.threads, e.run.threads);
					m.slot_entries.push_back(e);
					++ss.entry_count;
				} else if (step.kind == 1) {
					ResBatchEntry e{};
					e.sg = m.plan.segments[step.index];
					e.sg.pad = step.index;
					if (first) e.sg.has_prev = 0;  // a job starts from cost 0
					e.prev = lane.d_pr[c.flip];
					e.cur = lane.d_pr[c.flip ^ 1];
					e.score_out = (last && !job.final) ? m.d_job_scores + job_id : nullptr;
					ss.io[0] = lane.d_pr[c.flip]; ss.io[1] = lane.d_pr[c.flip ^ 1];
					const size_t lds = e.sg.kind == 1
						? ((((size_t)e.sg.ncols * (PED_LDSWORDS + PED_TABLE) + (size_t)e.sg.n_terms * 2 + 3) & ~(size_t)3) * 4 + 2 * ((size_t)16 << e.sg.max_l) + (size_t)e.sg.stage_words * 8)
						: ((size_t)e.sg.ncols * (64 + RES_TABLE) * 4 + 2 * ((size_t)4 << e.sg.max_l) + (size_t)e.sg.stage_words * 8);
					ss.lds = std::max(ss.lds, lds);
					ss.grid_x = std::max(ss.grid_x, 1u << (e.sg.g - e.sg.half));
					ss.threads = std::max(ss.threads, e.sg.threads);
					ss.sym = ss.sym || e.sg.half || e.sg.in_half || e.sg.mirror_out;
					m.entries.push_back(e);
					++ss.entry_count;
				} else {
					ss.singles.push_back(Impl::Single{(uint32_t)li, si, c.flip, first, (last && !job.final) ? (int32_t)job_id : -1});
					ss.io[0] = lane.d_pr[c.flip]; ss.io[1] = lane.d_pr[c.flip ^ 1];
				}
				c.flip ^= 1;
				if (last) { ++c.job_i; c.step_i = 0; } else ++c.step_i;
			}
			if (!ss.entry_count && ss.singles.empty()) break;
			m.max_grid_x = std::max(m.max_grid_x, ss.grid_x * std::max(1u, ss.entry_count));
			m.schedule.push_back(std::move(ss));
		}
		if (m.windowed) {
			// one lane, one job: super-step i is step i.  Pass 1 = the schedule as it is, plus the kept columns and the
			// walk of the newest window; then every older window again, newest first.
			if (m.schedule.size() != m.jobs[0].steps.size()) { msg = "internal error: windowed schedule"; return WHAMD_ERR_DEVICE; }
			const size_t nw = m.windows.size();
			for (size_t wi = 0; wi + 1 < nw; ++wi) m.schedule[m.windows[wi].step_hi - 1].ck_save = (int32_t)wi;
			m.schedule.back().bt_window = (int32_t)(nw - 1);
			for (size_t wi = nw - 1; wi-- > 0;) {
				const Impl::Window& win = m.windows[wi];
				for (uint32_t pos = win.step_lo; pos < win.step_hi; ++pos) {
					Impl::SuperStep again = m.schedule[pos];
					again.ck_save = -1;
					again.ck_load = (pos == win.step_lo && wi > 0) ? (int32_t)(wi - 1) : -1;
					again.bt_window = pos + 1 == win.step_hi ? (int32_t)wi : -1;
					m.schedule.push_back(std::move(again));
				}
			}
		}
		m.step_brief.clear();
		m.entry_brief.clear();
		if (m.use_slots) {
			m.step_brief.reserve(m.schedule.size());
			m.entry_brief.reserve(m.slot_entries.size());
			for (const Impl::SuperStep& ss : m.schedule) {
				m.step_brief.push_back(Impl::StepBrief{(uint32_t)m.entry_brief.size(), (uint32_t)ss.lds, (uint16_t)ss.entry_count, (uint8_t)(ss.singles.empty() ? 0 : 1), 0});
				for (uint32_t q = 0; q < ss.entry_count; ++q) {
					const SlotBatchEntry& he = m.slot_entries[ss.entry_off + q];
					Impl::EntryBrief eb{};
					eb.grid_x = (uint16_t)(1u << (he.run.g - he.run.half));
					eb.threads = (uint16_t)he.run.threads;
					eb.lds_x = (uint32_t)((size_t)2 * he.run.threads * (4u << he.run.lr));
					// kernel variants of a group launch: 0 single individual (four cells per thread); 1 .. 5 pedigree runs (TB, NF) = (2,2) (2,4) (4,2) (4,4) (2,16); 6 single, eight cells;
					// 7 trio on factorised lines; 8 / 9 X runs with four / eight cells (slot_groupx); 10 quartet on factorised lines
					if (m.splan.ped) eb.variant = eb.variant_x = (uint8_t)(he.ex.nf == (uint32_t)PSLOT_FACT4 ? 10 : he.ex.nf == (uint32_t)PSLOT_FACT ? 7 : (he.ex.nf == 16 ? 5 : 1 + (he.ex.tb == 4 ? 2 : 0) + (he.ex.nf == 4 ? 1 : 0)));
					else {
						eb.variant = (uint8_t)(he.run.lr == 3 ? 6 : 0);
						eb.variant_x = (he.run.yflags & 8u) ? (uint8_t)(he.run.lr == 3 ? 9 : 8) : eb.variant;
					}
					m.entry_brief.push_back(eb);
				}
			}
		}
		void* d_entries = nullptr;
		HIP_TRY(up(&d_entries, m.entries.data(), m.entries.size() * sizeof(ResBatchEntry)));
		m.d_entries = (ResBatchEntry*)d_entries;
		void* d_slot_entries = nullptr;
		m.slot_entries.resize(m.slot_entries.size() + 2);   // (two unused entries behind the last: a group launch warms 512 bytes behind its own entry, slot_runx_core)
		HIP_TRY(up(&d_slot_entries, m.slot_entries.data(), m.slot_entries.size() * sizeof(SlotBatchEntry)));
		m.slot_entries.resize(m.slot_entries.size() - 2);
		m.d_slot_entries = (SlotBatchEntry*)d_slot_entries;
		std::vector<BtJob> btjobs;
		for (const Impl::Job& job : m.jobs) btjobs.push_back(BtJob{job.unit_off, job.unit_count, job.final ? 1u : 0u, 0u});
		void* d_btjobs = nullptr;
		HIP_TRY(up(&d_btjobs, btjobs.data(), btjobs.size() * sizeof(BtJob)));
		m.d_btjobs = (BtJob*)d_btjobs;
	}
	HIP_TRY(flush_slab());   // (everything is staged; the table kernels below read it)
	m.dp.cols = m.d_cols;
	m.dp.term_ptr = (const uint32_t*)d_term_ptr;
	m.dp.terms = (const DevTerm*)d_terms;
	if (ped_slots) {
		// the cost-form tables of every run (slots.h), once per table: blockIdx.y = run
		uint32_t most = 0;
		for (size_t ri = 0; ri < m.splan.runs.size(); ++ri)
			most = std::max<uint32_t>(most, (m.splan.pextra[ri].fwn << m.splan.runs[ri].g) + (m.splan.pextra[ri].fwn << m.splan.runs[ri].lw) + m.splan.runs[ri].ncols * (64u * pslot_ns(m.splan.pextra[ri].nf) + p.T * pslot_nk(m.splan.pextra[ri].nf)));
		const uint32_t bx = std::max(1u, std::min(1024u, (most + 255u) / 256u));
		for (size_t r0 = 0; r0 < m.splan.runs.size(); r0 += 32768) {   // (gridDim.y <= 65535)
			const uint32_t ny = (uint32_t)std::min<size_t>(32768, m.splan.runs.size() - r0);
			hipLaunchKernelGGL(pedslot_tables, dim3(bx, ny), dim3(256), 0, us, m.dp, (const SlotRun*)d_pruns + r0, (const PedSlotExtra*)d_pextra + r0, (uint32_t*)d_ptab,
			                   (const DevTerm*)d_fterms);
		}
		HIP_TRY(hipGetLastError());
	}
	if (m.use_slots && !ped_slots && !m.splan.runs.empty()) {
		uint32_t most = 0;
		for (const SlotRun& run : m.splan.runs) most = std::max<uint32_t>(most, ((run.ncols + 8u) << (run.g - run.half)) + (run.ncols + 8u) * ((run.threads >> 6) + 64u));
		const uint32_t bx = std::max(1u, std::min(64u, (most + 255u) / 256u));
		for (size_t r0 = 0; r0 < m.splan.runs.size(); r0 += 32768) {
			const uint32_t ny = (uint32_t)std::min<size_t>(32768, m.splan.runs.size() - r0);
			hipLaunchKernelGGL(slot_tables, dim3(bx, ny), dim3(256), 0, us, m.dp, (const SlotRun*)d_sruns + r0, (uint32_t*)d_stab);
		}
		HIP_TRY(hipGetLastError());
	}
	// No host wait: the solve is ordered behind `ev_upload` on the device (begin_solve) and the staging area behind its own event (StageSession::park) -- a create
	// used to end with hipStreamSynchronize: 1.5 - 2 ms of copy tail and table kernel for configs[2], and under many concurrent creates every worker thread sat in
	// the queue of the others' copies (half of a create's wall time at 16 workers).  WHAMD_SYNC_UPLOAD=1 (debug library) restores the wait.
	HIP_TRY(hipEventRecord(m.ev_upload, us));
	m.upload_pending = true;
	if (debug_env("WHAMD_SYNC_UPLOAD") || unstaged_copies || !stage.image || !stage.park()) {
		HIP_TRY(hipStreamSynchronize(us));
		stage.finish();
	}
	m.dp.delta = (const int32_t*)d_delta;
	m.dp.term_ptr = (const uint32_t*)d_term_ptr;
	m.dp.terms = (const DevTerm*)d_terms;
	m.dp.segs = (const uint32_t*)d_segs;
	m.dp.bt = (uint8_t*)d_bt;
	m.dp.keys = (unsigned long long*)d_keys;
	m.dp.last_keys = (unsigned long long*)d_last_keys;
	m.dp.bt_state = m.windowed ? m.d_bt_state : nullptr;
	m.dp.res_cols = (const ResColumn*)d_rcol;
	m.dp.res_bt = (const ResBacktrace*)d_rbt;
	m.dp.dbg = nullptr;
	if (getenv("WHAMD_DEBUG_TIMING")) {
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		fprintf(stderr, "[whamd timing] upload: plan %.1f ms, descriptors + copies %.1f ms, backtrace arena (%.2f GB) %.1f ms, rest %.1f ms\n",
		        ms(tu0, tu1), ms(tu1, tu2), (double)bt / 1e9, ms(tu2, tu3), ms(tu3, std::chrono::steady_clock::now()));
	}
	if (debug_env("WHAMD_DEBUG_STAMPS") && !m.use_slots) {   // in-kernel cycle stamps of the LDS-resident runs (WHAMD_DEBUG_TIMING: host phases only)
		void* d_dbg = nullptr;
		const size_t dbg_bytes = (m.plan.segments.size() + 1) * 64 + 4 * 512 * 16 + 64;
		HIP_TRY(alloc(&d_dbg, dbg_bytes));
		HIP_TRY(hipMemset(d_dbg, 0, dbg_bytes));
		m.dp.dbg = (unsigned long long*)d_dbg;
		m.dp.dbg_wg_off = (uint32_t)((m.plan.segments.size() + 1) * 8);
		m.dp.dbg_flags = (uint32_t)atoi(debug_env("WHAMD_DEBUG_STAMPS"));
	}
	if (debug_env("WHAMD_SLOT_STAMPS") && m.use_slots) {   // in-kernel cycle stamps of workgroup 0 / wave 0 of every slot run
		void* d_dbg = nullptr;
		const size_t dbg_bytes = (m.splan.runs.size() + 1) * 48 * 8 + 4 * 512 * 16 + 128;   // + the backtrace kernel's own stamps
		HIP_TRY(alloc(&d_dbg, dbg_bytes));
		HIP_TRY(hipMemset(d_dbg, 0, dbg_bytes));
		m.dp.dbg = (unsigned long long*)d_dbg;
		m.dp.dbg_wg_off = (uint32_t)((m.splan.runs.size() + 1) * 48);
		for (size_t i = 0; i < m.slot_entries.size(); ++i) m.slot_entries[i].run.pad = (uint32_t)i;
	}
	if (const char* skip = debug_env("WHAMD_SLOT_SKIP")) m.dp.dbg_flags = (uint32_t)atoi(skip);  // timing experiments (results invalid): 1 no exit
	                                                                                          // stores, 2 no records, 4 one column per run, 8 no ending reads, 16 no cost update
	m.dp.n_cols = n;
	m.dp.T = p.T;
	m.dp.tbits = tbits;
	m.dp.n_ind = p.n_ind;
	// kernels with more than 64 KiB of dynamic LDS need the opt-in on every device they run on -- once per process and device (23 driver calls
	// per table were a tenth of a coverage-15 create when many tables are built at once)
	{
		static std::mutex attr_mu;
		static unsigned long long attr_done = 0;   // bit = device
		std::lock_guard<std::mutex> lock(attr_mu);
		if (device >= 64 || !((attr_done >> device) & 1ull)) {
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_segment<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_segment<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef WHAMD_DEBUG_BUILD
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_segment<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_batch<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_batch<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(backtrace_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_segment_ped<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef WHAMD_DEBUG_BUILD
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_segment_ped<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((resident_segment_ped<false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 24, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 24, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 32, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 32, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		// every instantiation launch_slot_run / enqueue_group may pick: a 9..16-column run of 512 threads needs slotx_lds_bytes(512, 16) = 72 KB, above the
		// 64 KB a kernel gets without the attribute (XC = 0: streamed operands, 16 KB; XC = 8: 56 KB -- registered all the same, the limit is per function)
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 0, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 0, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 8, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 8, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 16, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 16, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_groupx<2, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_groupx<3, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef WHAMD_DEBUG_BUILD
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 24, true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 24, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 32, true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((slot_runx<2, 32, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
#ifdef WHAMD_DEBUG_BUILD
#define WHAMD_PSLOTX_ATTR(TBV, NFV) \
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_runx<TBV, NFV, 32, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_runx<TBV, NFV, 32, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		WHAMD_PSLOTX_ATTR(2, 2) WHAMD_PSLOTX_ATTR(2, 4) WHAMD_PSLOTX_ATTR(4, 2) WHAMD_PSLOTX_ATTR(4, 4) WHAMD_PSLOTX_ATTR(2, 16) WHAMD_PSLOTX_ATTR(2, PSLOT_FACT)
#undef WHAMD_PSLOTX_ATTR
#endif
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, 4, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, 4, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, 4, true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, 4, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, 2, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, 2, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, 2, true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, 2, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, 4, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, 4, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, 4, true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, 4, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, 16, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, 16, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, 16, true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, 16, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_group<2, 16>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, PSLOT_FACT, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, PSLOT_FACT, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, PSLOT_FACT, true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<2, PSLOT_FACT, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_group<2, PSLOT_FACT>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, PSLOT_FACT4, false, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, PSLOT_FACT4, false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, PSLOT_FACT4, true, false>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_run<4, PSLOT_FACT4, true, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_group<4, PSLOT_FACT4>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_group<2, 2>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_group<2, 4>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_group<4, 2>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>((pedslot_group<4, 4>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
			if (device < 64) attr_done |= 1ull << device;
		}
	}
	m.d_bt_entry = nullptr;
	if (m.use_chunks) {   // (m.dp is complete here)
		BtGroupEntry& e = m.h_bt_entry;
		e = BtGroupEntry{};
		e.P = m.dp;
		e.units = m.d_units; e.chunks = m.d_chunks;
		e.n_chunks = (uint32_t)m.chunks.size(); e.n_units = (uint32_t)m.units.size(); e.n_orient_max = m.n_orient_max; e.n_cols = n;
		e.path2 = m.d_path2; e.trans2 = m.d_trans2; e.out_score = m.d_score; e.unit_x2 = m.d_unit_x; e.guess = m.d_guess; e.sel = m.d_sel; e.counters = m.d_bt_counters;
		e.path_index = m.d_path_index; e.path_trans = m.d_path_trans;
		if (m.device_superreads) e.super = m.super_args;
		void* d_entry = nullptr;
		HIP_TRY(alloc(&d_entry, sizeof(BtGroupEntry)));
		HIP_TRY(hipMemcpyAsync(d_entry, &m.h_bt_entry, sizeof(BtGroupEntry), hipMemcpyHostToDevice, us));   // (the source is a member: it outlives the copy)
		HIP_TRY(hipEventRecord(m.ev_upload, us));
		m.d_bt_entry = (BtGroupEntry*)d_entry;
	}
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
// One per-column step (column_step_fused, or column_step_keys + column_finalize) of a lane.
void DeviceTable::Impl::launch_column_step(const Problem& p, const Step& step, const Lane& lane, const uint32_t* prev, uint32_t* cur,
                                           uint64_t& launches) {
	Impl& m = *this;
	DevProblem dp = m.dp;
	dp.keys = lane.d_keys;
	const uint32_t c = step.index;
	const DevColumn& d = m.cols[c];
	if (d.mode == 0) {
		const uint32_t threads = 1u << d.f;
		const uint32_t block = std::min<uint32_t>(256, threads);
		hipLaunchKernelGGL(m.fused, dim3(threads / block), dim3(block), 0, m.run_stream, dp, c, prev, cur);
		launches += 1;
	} else {
		const uint64_t total = (1ull << (d.f + d.ebits - d.eloop)) * (m.wide ? p.T : 1u);
		const uint32_t block = (uint32_t)std::min<uint64_t>(256, (total + 63) / 64 * 64);
		if (m.wide) hipLaunchKernelGGL(column_step_wide, dim3((uint32_t)((total + block - 1) / block)), dim3(block), 0, m.run_stream, dp, c, prev, (uint32_t)total);
		else hipLaunchKernelGGL(m.keysfn, dim3((uint32_t)((total + block - 1) / block)), dim3(block), 0, m.run_stream, dp, c, prev, (uint32_t)total);
		const uint32_t entries = (1u << d.f) * p.T;
		const uint32_t fblock = std::min<uint32_t>(256, (entries + 63) / 64 * 64);
		hipLaunchKernelGGL(column_finalize, dim3((entries + fblock - 1) / fblock), dim3(fblock), 0, m.run_stream, dp, c, cur, entries);
		launches += 2;
	}
}

// One run as a launch of its own (kernel arguments by value).
void DeviceTable::Impl::launch_run(const ResBatchEntry& e, uint32_t step_index, uint64_t& launches) {
	Impl& m = *this;
	const ResSegment& sg = e.sg;
	if (sg.kind == 1) {
		const size_t words = ((size_t)sg.ncols * (PED_LDSWORDS + PED_TABLE) + (size_t)sg.n_terms * 2 + 3) & ~(size_t)3;
		const size_t lds_ped = words * 4 + 2 * ((size_t)16 << sg.max_l) + (size_t)sg.stage_words * 8;
#ifdef WHAMD_DEBUG_BUILD
		if (m.dp.dbg) hipLaunchKernelGGL(resident_segment_ped<true>, dim3(1u << sg.g), dim3(sg.threads), lds_ped, m.run_stream, m.dp, sg, e.prev, e.cur);
		else
#endif
		if (sg.in_mirror_bit && m.use_chunks) hipLaunchKernelGGL((resident_segment_ped<false, true>), dim3(1u << sg.g), dim3(sg.threads), lds_ped, m.run_stream, m.dp, sg, e.prev, e.cur);
		else hipLaunchKernelGGL(resident_segment_ped<false>, dim3(1u << sg.g), dim3(sg.threads), lds_ped, m.run_stream, m.dp, sg, e.prev, e.cur);
	} else {
		const size_t lds = (size_t)sg.ncols * (64 + RES_TABLE) * 4 + 2 * ((size_t)4 << sg.max_l) + (size_t)sg.stage_words * 8;
		const bool sym = sg.half || sg.in_half || sg.mirror_out;
		const dim3 grid(1u << (sg.g - sg.half)), block(sg.threads);
#ifdef WHAMD_DEBUG_BUILD
		if (m.dp.dbg) hipLaunchKernelGGL((resident_segment<true, true>), grid, block, lds, m.run_stream, m.dp, sg, e.prev, e.cur, e.score_out);
		else
#endif
		if (sym) hipLaunchKernelGGL((resident_segment<false, true>), grid, block, lds, m.run_stream, m.dp, sg, e.prev, e.cur, e.score_out);
		else hipLaunchKernelGGL((resident_segment<false, false>), grid, block, lds, m.run_stream, m.dp, sg, e.prev, e.cur, e.score_out);
	}
	launches += 1;
}

// One slot run as a launch of its own (kernel arguments by value).
void DeviceTable::Impl::launch_slot_run(const SlotBatchEntry& e, uint64_t& launches) {
	Impl& m = *this;
	const SlotRun& run = e.run;
	const bool spec = run.spec_id != 0 && m.use_chunks && !debug_env("WHAMD_NO_SPEC_KERNEL");
	if (m.splan.ped) {
		const PedSlotExtra& ex = m.splan.pextra[e.pad];
		const size_t lds_ped = pedslot_lds_bytes(run.threads, run.ncols, ex);
		const dim3 grid(1u << run.g), block(run.threads);
#ifdef WHAMD_DEBUG_BUILD
		if ((run.yflags & 8u) && !m.side_by_side) {   // X run (kernels_pedslots.h, pedslot_runx; WHAMD_PED_XRUN=1): the costs of every column formed in the prologue, scalars through the scalar cache
#define WHAMD_PSLOTX_LAUNCH2(TBV, NFV, XCV) do { const size_t lds_x = pedslotx_lds_bytes(run.threads, XCV); \
		if (spec) hipLaunchKernelGGL((pedslot_runx<TBV, NFV, XCV, true>), grid, block, lds_x, m.run_stream, m.dp, run, ex, e.prev, e.cur); \
		else hipLaunchKernelGGL((pedslot_runx<TBV, NFV, XCV, false>), grid, block, lds_x, m.run_stream, m.dp, run, ex, e.prev, e.cur); } while (0)
#define WHAMD_PSLOTX_LAUNCH(TBV, NFV) do { if (run.ncols <= 16u) WHAMD_PSLOTX_LAUNCH2(TBV, NFV, 16); else WHAMD_PSLOTX_LAUNCH2(TBV, NFV, 32); } while (0)
			if (ex.tb == 2 && ex.nf == (uint32_t)PSLOT_FACT) WHAMD_PSLOTX_LAUNCH(2, PSLOT_FACT);
			else if (ex.tb == 2 && ex.nf == 16) WHAMD_PSLOTX_LAUNCH(2, 16);
			else if (ex.tb == 2 && ex.nf == 2) WHAMD_PSLOTX_LAUNCH(2, 2);
			else if (ex.tb == 2) WHAMD_PSLOTX_LAUNCH(2, 4);
			else if (ex.nf == 2) WHAMD_PSLOTX_LAUNCH(4, 2);
			else WHAMD_PSLOTX_LAUNCH(4, 4);
#undef WHAMD_PSLOTX_LAUNCH
#undef WHAMD_PSLOTX_LAUNCH2
			launches += 1;
			return;
		}
#endif
#define WHAMD_PSLOT_LAUNCH(TBV, NFV, SPECV) do { if (run.yflags & 16u) hipLaunchKernelGGL((pedslot_run<TBV, NFV, SPECV, true>), grid, block, lds_ped, m.run_stream, m.dp, run, ex, e.prev, e.cur); \
		else hipLaunchKernelGGL((pedslot_run<TBV, NFV, SPECV, false>), grid, block, lds_ped, m.run_stream, m.dp, run, ex, e.prev, e.cur); } while (0)
		if (ex.tb == 2 && ex.nf == (uint32_t)PSLOT_FACT) { if (spec) WHAMD_PSLOT_LAUNCH(2, PSLOT_FACT, true); else WHAMD_PSLOT_LAUNCH(2, PSLOT_FACT, false); }
		else if (ex.tb == 4 && ex.nf == (uint32_t)PSLOT_FACT4) { if (spec) WHAMD_PSLOT_LAUNCH(4, PSLOT_FACT4, true); else WHAMD_PSLOT_LAUNCH(4, PSLOT_FACT4, false); }
		else if (ex.tb == 2 && ex.nf == 16) { if (spec) WHAMD_PSLOT_LAUNCH(2, 16, true); else WHAMD_PSLOT_LAUNCH(2, 16, false); }
		else if (ex.tb == 2 && ex.nf == 2) { if (spec) WHAMD_PSLOT_LAUNCH(2, 2, true); else WHAMD_PSLOT_LAUNCH(2, 2, false); }
		else if (ex.tb == 2) { if (spec) WHAMD_PSLOT_LAUNCH(2, 4, true); else WHAMD_PSLOT_LAUNCH(2, 4, false); }
		else if (ex.nf == 2) { if (spec) WHAMD_PSLOT_LAUNCH(4, 2, true); else WHAMD_PSLOT_LAUNCH(4, 2, false); }
		else { if (spec) WHAMD_PSLOT_LAUNCH(4, 4, true); else WHAMD_PSLOT_LAUNCH(4, 4, false); }
#undef WHAMD_PSLOT_LAUNCH
		launches += 1;
		return;
	}
	const size_t lds = slot_run_lds_bytes(run.threads, run.lr, run.ncols);   // wave-slot exchange + hot lines + per-wave A + lane sums
	const dim3 grid(1u << (run.g - run.half)), block(run.threads);
	if ((run.yflags & 8u) && run.lr == 2u) {   // X run with four cells per thread: registers instead of LDS lines (LDS: the wave-slot exchange buffers + the threads' own operand lines)
		// (no LDS lines: room for other tables' workgroups on the CU -- or, from 1 024 workgroups on, for FOUR of this table's own instead of two:
		// a launch of 2 048 workgroups -- coverage 23 -- takes 36.2 instead of 56.2 us, scripts/gpu_wide_ab.py)
		const bool streamed = m.side_by_side || debug_env("WHAMD_XSTREAM") || (grid.x >= 1024u && !debug_env("WHAMD_NO_WIDE_LAYOUT"));
		const size_t lds_x = streamed ? (size_t)2 * run.threads * 16 : slotx_lds_bytes(run.threads, (run.ncols + 7u) & ~7u);
		// narrow tables: the run's workgroups packed onto one XCD (slot_runx: eight times the grid, every eighth workgroup works)
		const uint32_t pack = (grid.x <= 32u && !m.side_by_side && !debug_env("WHAMD_NO_XCD_PACK")) ? 1u : 0u;
		const dim3 xgrid(pack ? grid.x * 8u : grid.x);
#define WHAMD_SLOTX_LAUNCH(XCV, DBGV, SPECV) hipLaunchKernelGGL((slot_runx<2, XCV, DBGV, SPECV>), xgrid, block, lds_x, m.run_stream, m.dp, run, e.prev, e.cur, e.score_out, pack)
		if (streamed) {
			if (spec) WHAMD_SLOTX_LAUNCH(0, false, true); else WHAMD_SLOTX_LAUNCH(0, false, false);
		} else
#ifdef WHAMD_DEBUG_BUILD
		if (m.dp.dbg != nullptr || m.dp.dbg_flags != 0) {
			if (run.ncols <= 24u) { if (spec) WHAMD_SLOTX_LAUNCH(24, true, true); else WHAMD_SLOTX_LAUNCH(24, true, false); }
			else { if (spec) WHAMD_SLOTX_LAUNCH(32, true, true); else WHAMD_SLOTX_LAUNCH(32, true, false); }
		} else
#endif
		if (run.ncols <= 8u) { if (spec) WHAMD_SLOTX_LAUNCH(8, false, true); else WHAMD_SLOTX_LAUNCH(8, false, false); }       // (the prologue forms the operands of XC columns:
		else if (run.ncols <= 16u) { if (spec) WHAMD_SLOTX_LAUNCH(16, false, true); else WHAMD_SLOTX_LAUNCH(16, false, false); }   //  an irregular layout's runs are ~10 columns long)
		else if (run.ncols <= 24u) { if (spec) WHAMD_SLOTX_LAUNCH(24, false, true); else WHAMD_SLOTX_LAUNCH(24, false, false); }
		else { if (spec) WHAMD_SLOTX_LAUNCH(32, false, true); else WHAMD_SLOTX_LAUNCH(32, false, false); }
#undef WHAMD_SLOTX_LAUNCH
		launches += 1;
		return;
	}
	// (the instantiations with cycle stamps and timing switches exist in the debug library only: debug_build.h)
#ifdef WHAMD_DEBUG_BUILD
	const bool dbg = m.dp.dbg != nullptr || m.dp.dbg_flags != 0;
#define WHAMD_IF_DBG(stmt) if (dbg) { stmt; } else
#else
#define WHAMD_IF_DBG(stmt)
#endif
#define WHAMD_SLOT_LAUNCH(LRV, DBGV, SPECV) hipLaunchKernelGGL((slot_run<LRV, DBGV, SPECV>), grid, block, lds, m.run_stream, m.dp, run, e.prev, e.cur, e.score_out)
	if (run.lr == 3 && (run.yflags & 1u)) {   // Y-form run, eight cells per thread
#define WHAMD_SLOT_LAUNCH_Y3(DBGV, SPECV) hipLaunchKernelGGL((slot_run<3, DBGV, SPECV, true>), grid, block, lds, m.run_stream, m.dp, run, e.prev, e.cur, e.score_out)
		WHAMD_IF_DBG(if (spec) WHAMD_SLOT_LAUNCH_Y3(true, true); else WHAMD_SLOT_LAUNCH_Y3(true, false)) { if (spec) WHAMD_SLOT_LAUNCH_Y3(false, true); else WHAMD_SLOT_LAUNCH_Y3(false, false); }
#undef WHAMD_SLOT_LAUNCH_Y3
	} else if (run.lr == 3) {
		WHAMD_IF_DBG(if (spec) WHAMD_SLOT_LAUNCH(3, true, true); else WHAMD_SLOT_LAUNCH(3, true, false)) { if (spec) WHAMD_SLOT_LAUNCH(3, false, true); else WHAMD_SLOT_LAUNCH(3, false, false); }
	} else if (run.lr == 1) {
		WHAMD_IF_DBG(if (spec) WHAMD_SLOT_LAUNCH(1, true, true); else WHAMD_SLOT_LAUNCH(1, true, false)) { if (spec) WHAMD_SLOT_LAUNCH(1, false, true); else WHAMD_SLOT_LAUNCH(1, false, false); }
	} else if (run.yflags & 1u) {   // Y-form run (slot_plan.cpp): one instruction per cell-column
#define WHAMD_SLOT_LAUNCH_Y(DBGV, SPECV) hipLaunchKernelGGL((slot_run<2, DBGV, SPECV, true>), grid, block, lds, m.run_stream, m.dp, run, e.prev, e.cur, e.score_out)
		WHAMD_IF_DBG(if (spec) WHAMD_SLOT_LAUNCH_Y(true, true); else WHAMD_SLOT_LAUNCH_Y(true, false)) { if (spec) WHAMD_SLOT_LAUNCH_Y(false, true); else WHAMD_SLOT_LAUNCH_Y(false, false); }
#undef WHAMD_SLOT_LAUNCH_Y
	} else {
		WHAMD_IF_DBG(if (spec) WHAMD_SLOT_LAUNCH(2, true, true); else WHAMD_SLOT_LAUNCH(2, true, false)) { if (spec) WHAMD_SLOT_LAUNCH(2, false, true); else WHAMD_SLOT_LAUNCH(2, false, false); }
	}
#undef WHAMD_SLOT_LAUNCH
#undef WHAMD_IF_DBG
	launches += 1;
}

// Resumable submission: the first call does the preamble, every call submits super-steps until at least `budget`
// launches went out, the call that runs out of super-steps appends the backtrace and the downloads.  Lets one host
// thread interleave the launch sequences of several tables (whamd_dptable_enqueue_many).
whamd_status_t DeviceTable::enqueue_some(const Problem& p, Solution& s, uint64_t budget, bool& done, std::string& msg) {
	const whamd_status_t status = enqueue_some_unguarded(p, s, budget, done, msg);
	if (status != WHAMD_OK) abort_enqueue();  // never leave a half-submitted schedule behind: the next enqueue starts over
	return status;
}

// Drops a partially submitted solve: waits for what is already on the stream and rewinds the resumable cursor, so that
// a later enqueue()/solve() of this table begins with the preamble again (key re-arm, events, path buffers).
void DeviceTable::abort_enqueue() {
	Impl& m = *impl_;
	if (m.stream) {
		(void)hipSetDevice(m.device);
		(void)hipStreamSynchronize(m.stream);
		(void)hipGetLastError();
	}
	m.enqueue_open = false;
	m.next_super = 0;
}

// The preamble of a solve on m.run_stream: path buffers, key re-arm, the start event, the lookup tables of the LDS-resident paths.
whamd_status_t DeviceTable::Impl::begin_solve(const Problem& p, Solution& s, std::string& msg) {
	Impl& m = *this;
	const uint32_t n = p.n_cols;
	s.path_index.assign(n, 0);
	s.path_trans.assign(n, 0);
	s.superreads_done = false;
	m.launches = 0;
	m.next_super = 0;
	if (n == 0) return WHAMD_OK;
	if (m.run_stream == m.stream) m.own_stream_used = true;
	if (m.ev_upload && (m.upload_pending || m.run_stream != m.stream)) HIP_TRY(hipStreamWaitEvent(m.run_stream, m.ev_upload, 0));   // (the uploads went through an upload stream; once a solve has been collected they are known to be there)
	for (const Impl::Lane& lane : m.lanes) HIP_TRY(hipMemsetAsync(lane.d_keys, 0xFF, m.key_entries * 8, m.run_stream));
	HIP_TRY(hipMemsetAsync(m.dp.last_keys, 0xFF, (size_t)MAX_T_WIDE * 8, m.run_stream));
	if (m.use_chunks) HIP_TRY(hipMemsetAsync(m.dp.spec_keys, 0xFF, ((size_t)m.n_spec + 1) * m.dp.spec_stride * 8, m.run_stream));
	if (m.windowed) HIP_TRY(hipMemsetAsync(m.d_path_trans, 0, (size_t)n * 4, m.run_stream));
	m.timing_pending = false;   // (the events are this solve's from here on)
	HIP_TRY(hipEventRecord(m.ev0, m.run_stream));
	if (!m.plan.ped_columns.empty()) {
		const uint32_t entries = (uint32_t)m.plan.ped_columns.size() * PED_TABLE;
		hipLaunchKernelGGL(ped_tables, dim3((entries + 255) / 256), dim3(256), 0, m.run_stream, m.dp.ped_cols, (uint32_t)m.plan.ped_columns.size(), m.dp.ped_tables);
	} else if (!m.use_slots && !m.plan.columns.empty()) {
		const uint32_t entries = (uint32_t)m.plan.columns.size() * RES_TABLE;
		hipLaunchKernelGGL(resident_tables, dim3((entries + 255) / 256), dim3(256), 0, m.run_stream, m.dp.res_cols, (uint32_t)m.plan.columns.size(), m.dp.res_tables);
	}
	return WHAMD_OK;
}

// The per-column steps of one super-step (each a launch of its own on m.run_stream).
whamd_status_t DeviceTable::Impl::submit_singles(const Problem& p, const SuperStep& ss, uint64_t& launches, std::string& msg) {
	Impl& m = *this;
	for (const Impl::Single& sg : ss.singles) {
		const Impl::Lane& lane = m.lanes[sg.lane];
		if (sg.zero_prev) HIP_TRY(hipMemsetAsync(lane.d_pr[sg.flip], 0, 4 * (size_t)p.T, m.run_stream));
		m.launch_column_step(p, m.plan.steps[sg.step], lane, lane.d_pr[sg.flip], lane.d_pr[sg.flip ^ 1], launches);
		if (sg.score_job >= 0)
			HIP_TRY(hipMemcpyAsync(m.d_job_scores + sg.score_job, lane.d_pr[sg.flip ^ 1], 4, hipMemcpyDeviceToDevice, m.run_stream));
	}
	return WHAMD_OK;
}

// Everything after the forward pass, on the table's OWN stream: backtrace, downloads, events.
whamd_status_t DeviceTable::Impl::submit_tail(const Problem& p, std::string& msg, hipStream_t tail_stream, bool backtrace_done, bool superreads_done) {
	Impl& m = *this;
	const hipStream_t ts = tail_stream ? tail_stream : m.stream;
	m.tail_elsewhere = ts != m.stream;
	if (ts == m.stream) m.own_stream_used = true;
	m.tail_stream = ts;
	{
		static std::atomic<uint64_t> seq{0};
		m.tail_seq = ++seq;
	}
	const uint32_t n = p.n_cols;
	HIP_TRY(hipGetLastError());
	if (!backtrace_done) HIP_TRY(hipEventRecord(m.ev1, ts));   // (backtrace_done: a batched launch walked this table with the rest of its group; ev1 was recorded in front of it)
	if (backtrace_done) {
	} else if (m.use_chunks) {
		hipLaunchKernelGGL(backtrace_chunks, dim3(m.n_orient_max * (uint32_t)m.chunks.size()), dim3(256), m.chunk_lds, ts, m.dp, m.d_units, m.d_chunks,
		                   (uint32_t)m.chunks.size(), (uint32_t)m.units.size(), 0u, m.n_orient_max, m.d_path2, m.d_trans2, m.d_score, m.d_unit_x, m.d_guess, m.d_sel, m.d_bt_counters);
		hipLaunchKernelGGL(backtrace_chunks, dim3(1), dim3(256), m.chunk_lds, ts, m.dp, m.d_units, m.d_chunks,
		                   (uint32_t)m.chunks.size(), (uint32_t)m.units.size(), 1u, m.n_orient_max, m.d_path2, m.d_trans2, m.d_score, m.d_unit_x, m.d_guess, m.d_sel, m.d_bt_counters);
		hipLaunchKernelGGL(backtrace_gather, dim3((uint32_t)m.units.size()), dim3(64), 0, ts, m.d_units, (uint32_t)m.units.size(), n, m.d_path2, m.d_trans2, m.d_sel,
		                   m.d_path_index, m.d_path_trans);
	} else if (!m.windowed)   // (windowed: every window was walked right after its steps)
	hipLaunchKernelGGL(backtrace_kernel, dim3((uint32_t)m.jobs.size()), dim3(1024), m.bt_lds, ts, m.dp, m.d_units, m.d_btjobs,
	                   m.d_path_index, m.d_path_trans, m.d_score);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(m.ev2, ts));
	if (m.device_superreads) {
		if (!superreads_done) {
			hipLaunchKernelGGL(superreads_single, dim3((n + 255u) / 256u), dim3(256), 0, ts, m.super_args);
			HIP_TRY(hipGetLastError());
		}
	}
	// ONE download per table (the device block has the pinned buffer's layout, upload()); pinned: a copy into pageable memory would block this call until the stream drains
	HIP_TRY(hipMemcpyAsync(m.h_pinned, m.d_path_index, (m.super_off + (m.device_superreads ? m.super_words : 0)) * sizeof(uint32_t), hipMemcpyDeviceToHost, ts));
	HIP_TRY(hipEventRecord(m.ev3, ts));
	return WHAMD_OK;
}

whamd_status_t DeviceTable::enqueue_some_unguarded(const Problem& p, Solution& s, uint64_t budget, bool& done, std::string& msg) {
	Impl& m = *impl_;
	const uint32_t n = p.n_cols;
	done = false;
	if (n) HIP_TRY(hipSetDevice(m.device));
	if (!m.enqueue_open) {
		m.enqueue_open = true;
		m.run_stream = m.stream;
		m.group_tables = 1;
		const whamd_status_t st = m.begin_solve(p, s, msg);
		if (st != WHAMD_OK) return st;
		if (n == 0) {  // src/pedigreedptable.cpp:88-92
			s.optimal_score = 0;
			m.enqueue_open = false;
			done = true;
			return WHAMD_OK;
		}
	}
	uint64_t launches = 0;
	while (m.next_super < m.schedule.size() && launches < budget) {
		const Impl::SuperStep& ss = m.schedule[m.next_super++];
		if (ss.ck_load >= 0) HIP_TRY(hipMemcpyAsync(ss.io[0], m.d_checkpoints + (size_t)ss.ck_load * m.checkpoint_bytes, m.checkpoint_bytes, hipMemcpyDeviceToDevice, m.stream));
		if (m.use_slots) {
			if (ss.entry_count == 1) m.launch_slot_run(m.slot_entries[ss.entry_off], launches);
			else if (ss.entry_count > 1) {
				if (m.slot_lr_used == 1) hipLaunchKernelGGL(slot_batch<1>, dim3(ss.grid_x, ss.entry_count), dim3(ss.threads), ss.lds, m.stream, m.dp, m.d_slot_entries + ss.entry_off);
				else if (m.slot_lr_used == 3) hipLaunchKernelGGL(slot_batch<3>, dim3(ss.grid_x, ss.entry_count), dim3(ss.threads), ss.lds, m.stream, m.dp, m.d_slot_entries + ss.entry_off);
				else hipLaunchKernelGGL(slot_batch<2>, dim3(ss.grid_x, ss.entry_count), dim3(ss.threads), ss.lds, m.stream, m.dp, m.d_slot_entries + ss.entry_off);
				launches += 1;
			}
		} else if (ss.entry_count == 1) {
			m.launch_run(m.entries[ss.entry_off], 0, launches);
		} else if (ss.entry_count > 1) {
			if (ss.sym) hipLaunchKernelGGL(resident_batch<true>, dim3(ss.grid_x, ss.entry_count), dim3(ss.threads), ss.lds, m.stream, m.dp, m.d_entries + ss.entry_off);
			else hipLaunchKernelGGL(resident_batch<false>, dim3(ss.grid_x, ss.entry_count), dim3(ss.threads), ss.lds, m.stream, m.dp, m.d_entries + ss.entry_off);
			launches += 1;
		}
		const whamd_status_t st = m.submit_singles(p, ss, launches, msg);
		if (st != WHAMD_OK) return st;
		if (ss.ck_save >= 0) HIP_TRY(hipMemcpyAsync(m.d_checkpoints + (size_t)ss.ck_save * m.checkpoint_bytes, ss.io[1], m.checkpoint_bytes, hipMemcpyDeviceToDevice, m.stream));
		if (ss.bt_window >= 0)
			hipLaunchKernelGGL(backtrace_kernel, dim3(1), dim3(1024), m.bt_lds, m.stream, m.dp, m.d_units, m.d_window_jobs + ss.bt_window,
			                   m.d_path_index, m.d_path_trans, m.d_score);
	}
	m.launches += launches;
	if (m.next_super < m.schedule.size()) return WHAMD_OK;
	const whamd_status_t st = m.submit_tail(p, msg);
	if (st != WHAMD_OK) return st;
	m.enqueue_open = false;
	done = true;
	return WHAMD_OK;
}

// ---------------------------------------------------------------------------------------------- group solve
// Several independent tables as ONE sequence of launches (whamd_dptable_enqueue_many): super-step k of the group = step k of every
// member that still has one; the runs of all members go out as one slot_group / pedslot_group launch per kernel variant
// (blockIdx.y = member), on the stream of the group's first table; per-column steps follow as launches of their own.  When the
// forward pass is submitted every member's own stream waits for it (one event) and runs that table's backtrace and downloads --
// those overlap across the members -- so whamd_dptable_wait works per table as before.
bool DeviceTable::group_eligible(const Problem& p) const {
	const Impl& m = *impl_;
	if (p.n_cols == 0 || !m.use_slots || m.windowed || m.enqueue_open || m.dp.dbg || (m.dp.dbg_flags && m.splan.ped)) return false;
	if (!m.splan.ped && m.slot_lr_used != 2 && m.slot_lr_used != 3) return false;   // (group kernels: four or eight cells per thread)
	return debug_env("WHAMD_NO_GROUP") == nullptr;
}

int DeviceTable::device_index() const { return impl_->device; }
uint32_t DeviceTable::widest_launch() const { return impl_->max_grid_x; }

whamd_status_t DeviceTable::enqueue_group(DeviceTable* const* tables, const Problem* const* problems, Solution* const* solutions, size_t n_tables, std::string& msg) {
	if (n_tables == 0) return WHAMD_OK;
	HIP_TRY(hipSetDevice(tables[0]->impl_->device));
	// ---- parts.  Tables that advance in lockstep are all in the same phase at the same time: every workgroup waits in its prologue
	// together, then they all compete for the issue slots together -- three full-width tables in ONE launch per super-step take 17 us
	// where three tables on their own streams, drifting against each other, take 11 (whamd_dptable_enqueue_many therefore keeps up to
	// four full-width tables on their own streams).  A group can be cut into PARTS, each a group of its own on the stream of its first
	// table, submitted round robin; measured on 24 full-width tables that is no gain over one part (6.30 M columns/s with 1 part,
	// 6.23 / 6.15 / 5.75 / 6.06 / 5.98 with 2 / 3 / 4 / 6 / 8: profiles/r04/), so one part is the default and WHAMD_GROUP_PARTS the experiment.
	uint64_t width = 0;
	for (size_t i = 0; i < n_tables; ++i) width += tables[i]->impl_->max_grid_x;
	size_t n_parts = 1;
	if (const char* e = debug_env("WHAMD_GROUP_PARTS")) n_parts = (size_t)std::max(1, atoi(e));
	n_parts = std::min(n_parts, n_tables);
	const bool tight = width > 768 && !debug_env("WHAMD_GROUP_LOOSE");   // more than three workgroups per CU: the variants held to 80 SGPRs (four workgroups per CU)
	struct Batch { SlotGroupArgs args; uint32_t grid_x = 0, threads = 0; size_t lds = 0; };
	struct Part {
		std::vector<size_t> members;   // positions in `tables`
		Impl* lead = nullptr;
		std::vector<Batch> batches;    // per kernel variant: 0 single individual (four cells per thread); 1 .. 5 pedigree runs (TB, NF) = (2,2) (2,4) (4,2) (4,4) (2,16)
	};
	std::vector<Part> parts(n_parts);
	for (size_t i = 0; i < n_tables; ++i) parts[i % n_parts].members.push_back(i);
	constexpr int NV = 11;  // kernel variants (6: single individual, eight cells per thread; 7: trio, factorised lines; 8 / 9: X runs of a single individual with four / eight cells per thread, slot_groupx; 10: quartet, factorised lines)
	for (Part& part : parts) { part.lead = tables[part.members[0]]->impl_; part.batches.resize(NV); }
	auto abort_all = [&]() {
		for (Part& part : parts) (void)hipStreamSynchronize(part.lead->stream);
		(void)hipGetLastError();
		for (size_t i = 0; i < n_tables; ++i) {
			Impl& m = *tables[i]->impl_;
			(void)hipStreamSynchronize(m.stream);
			m.enqueue_open = false;
			m.next_super = 0;
			m.run_stream = m.stream;
		}
	};
	size_t max_steps = 0;
	for (Part& part : parts)
		for (size_t i : part.members) {
			Impl& m = *tables[i]->impl_;
			m.enqueue_open = true;
			m.run_stream = part.lead->stream;
			m.group_tables = (uint32_t)part.members.size();
			const whamd_status_t st = m.begin_solve(*problems[i], *solutions[i], msg);
			if (st != WHAMD_OK) { abort_all(); return st; }
			max_steps = std::max(max_steps, m.schedule.size());
		}
	auto flush = [&](Part& part, int variant) {
		Batch& b = part.batches[variant];
		if (!b.args.n) return;
		// a table's workgroups on ONE XCD (slot_group_who): the table is the fast grid dimension, padded to a multiple of eight -- where that spreads the tables
		// evenly over the eight XCDs (a multiple of eight of them, or so many that the remainder does not matter)
		const bool by_table = (b.args.n % 8u == 0u || b.args.n >= 40u) && !debug_env("WHAMD_GROUP_BY_WORKGROUP");
		b.args.pad = by_table ? 1u : 0u;
		const dim3 grid = by_table ? dim3((b.args.n + 7u) & ~7u, b.grid_x) : dim3(b.grid_x, b.args.n), block(b.threads);
		hipStream_t stream = part.lead->stream;
		switch (variant) {
			case 0:
#ifdef WHAMD_DEBUG_BUILD
				if (part.lead->dp.dbg_flags) hipLaunchKernelGGL((slot_group<2, true, false>), grid, block, b.lds, stream, b.args);
				else
#endif
				if (tight) hipLaunchKernelGGL((slot_group<2, false, true>), grid, block, b.lds, stream, b.args);
				else hipLaunchKernelGGL((slot_group<2, false, false>), grid, block, b.lds, stream, b.args);
				break;
			case 1: hipLaunchKernelGGL((pedslot_group<2, 2>), grid, block, b.lds, stream, b.args); break;
			case 2: hipLaunchKernelGGL((pedslot_group<2, 4>), grid, block, b.lds, stream, b.args); break;
			case 3: hipLaunchKernelGGL((pedslot_group<4, 2>), grid, block, b.lds, stream, b.args); break;
			case 4: hipLaunchKernelGGL((pedslot_group<4, 4>), grid, block, b.lds, stream, b.args); break;
			case 5: hipLaunchKernelGGL((pedslot_group<2, 16>), grid, block, b.lds, stream, b.args); break;
			case 7: hipLaunchKernelGGL((pedslot_group<2, PSLOT_FACT>), grid, block, b.lds, stream, b.args); break;
			case 10: hipLaunchKernelGGL((pedslot_group<4, PSLOT_FACT4>), grid, block, b.lds, stream, b.args); break;
			case 8: hipLaunchKernelGGL((slot_groupx<2, false>), grid, block, b.lds, stream, b.args); break;
			case 9: hipLaunchKernelGGL((slot_groupx<3, false>), grid, block, b.lds, stream, b.args); break;
			default:
				if (tight) hipLaunchKernelGGL((slot_group<3, false, true>), grid, block, b.lds, stream, b.args);
				else hipLaunchKernelGGL((slot_group<3, false, false>), grid, block, b.lds, stream, b.args);
				break;
		}
		b.args.n = 0;
		b.grid_x = b.threads = 0;
		b.lds = 0;
	};
	std::vector<uint64_t> table_launches(n_tables, 0);
	std::vector<uint8_t> counted(n_tables * NV, 0);
	const auto t_submit0 = std::chrono::steady_clock::now();
#ifdef WHAMD_DEBUG_BUILD
	const bool touch_fat = debug_env("WHAMD_GROUP_TOUCH_ENTRIES") != nullptr;
	volatile uint64_t fat_sink = 0;
#endif
	for (size_t k = 0; k < max_steps; ++k) {
		for (Part& part : parts) {
			for (size_t i : part.members) std::fill(counted.begin() + i * NV, counted.begin() + i * NV + NV, 0);
			for (size_t i : part.members) {
				Impl& m = *tables[i]->impl_;
				if (k >= m.step_brief.size()) continue;
				const Impl::StepBrief& sb = m.step_brief[k];   // (a few bytes per table and super-step: Impl::StepBrief)
#ifdef WHAMD_DEBUG_BUILD
				if (touch_fat) {   // (A/B of the round-5 change: read what the loop used to read -- the super-step and its 320-byte entries)
					const Impl::SuperStep& ss = m.schedule[k];
					for (uint32_t q = 0; q < ss.entry_count; ++q) { const SlotBatchEntry& he = m.slot_entries[ss.entry_off + q]; fat_sink += he.run.yflags + he.run.g + he.run.threads + he.ex.nf + (uint32_t)ss.lds; }
				}
#endif
				for (uint32_t q = 0; q < sb.entry_count; ++q) {
					const Impl::EntryBrief& eb = m.entry_brief[sb.entry_off + q];
					const bool xrun = eb.variant_x != eb.variant && !m.dp.dbg_flags;   // (the X kernel: operands streamed from the tables, 16 KB of LDS)
					const int variant = xrun ? eb.variant_x : eb.variant;
					Batch& b = part.batches[variant];
					if (b.args.n == (uint32_t)SLOT_GROUP_MAX) {
						flush(part, variant);
						for (size_t j : part.members) if (counted[j * NV + variant]) { table_launches[j] += 1; counted[j * NV + variant] = 0; }
					}
					b.args.entry[b.args.n++] = m.d_slot_entries + sb.entry_off + q;
					b.grid_x = std::max<uint32_t>(b.grid_x, eb.grid_x);
					b.threads = std::max<uint32_t>(b.threads, eb.threads);
					b.lds = std::max<size_t>(b.lds, xrun ? eb.lds_x : sb.lds);
					counted[i * NV + variant] = 1;
				}
			}
			for (int v = 0; v < NV; ++v) {
				flush(part, v);
				for (size_t j : part.members) if (counted[j * NV + v]) table_launches[j] += 1;
			}
			for (size_t i : part.members) {
				Impl& m = *tables[i]->impl_;
				if (k >= m.step_brief.size() || !m.step_brief[k].has_singles) continue;
				const whamd_status_t st = m.submit_singles(*problems[i], m.schedule[k], table_launches[i], msg);
				if (st != WHAMD_OK) { abort_all(); return st; }
			}
		}
	}
	if (getenv("WHAMD_DEBUG_TIMING"))
		fprintf(stderr, "[whamd timing] group of %zu tables in %zu part(s): %zu super-steps submitted in %.2f ms (host)\n", n_tables, n_parts, max_steps,
		        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_submit0).count());
	if (hipGetLastError() != hipSuccess) { msg = "group launch failed"; abort_all(); return WHAMD_ERR_DEVICE; }
	for (Part& part : parts) {
		if (hipEventRecord(part.lead->ev_group, part.lead->stream) != hipSuccess) { msg = "hipEventRecord failed"; abort_all(); return WHAMD_ERR_DEVICE; }
		// the chunked backtrace of the part's members as ONE launch per mode + one gather (blockIdx.y = member), on the lead's stream
		std::vector<uint8_t> walked(n_tables, 0);
		if (!debug_env("WHAMD_TAIL_OWN_STREAM") && !debug_env("WHAMD_NO_GROUP_BACKTRACE")) {
			std::vector<size_t> batch;
			auto flush_bt = [&]() -> bool {
				if (batch.size() < 2) { batch.clear(); return true; }
				BtGroupArgs args{};
				uint32_t gx = 1, gu = 1;
				size_t lds = 0;
				for (size_t i : batch) {
					Impl& m = *tables[i]->impl_;
					args.entry[args.n++] = m.d_bt_entry;
					gx = std::max(gx, m.n_orient_max * (uint32_t)m.chunks.size());
					gu = std::max(gu, (uint32_t)m.units.size());
					lds = std::max(lds, m.chunk_lds);
					if (hipEventRecord(m.ev1, part.lead->stream) != hipSuccess) return false;
				}
				hipLaunchKernelGGL(backtrace_chunks_group, dim3(gx, args.n), dim3(256), lds, part.lead->stream, args, 0u);
				hipLaunchKernelGGL(backtrace_chunks_group, dim3(1, args.n), dim3(256), lds, part.lead->stream, args, 1u);
				hipLaunchKernelGGL(backtrace_gather_group, dim3(gu, args.n), dim3(64), 0, part.lead->stream, args);
				uint32_t gs = 0;   // (the superreads of the members the device makes them for, behind the gather)
				for (size_t i : batch) if (tables[i]->impl_->device_superreads) gs = std::max(gs, (problems[i]->n_cols + 255u) / 256u);
				if (gs) hipLaunchKernelGGL(superreads_group, dim3(gs, args.n), dim3(256), 0, part.lead->stream, args);
				if (hipGetLastError() != hipSuccess) return false;
				for (size_t i : batch) walked[i] = 1;
				batch.clear();
				return true;
			};
			for (size_t i : part.members) {
				Impl& m = *tables[i]->impl_;
				if (!m.use_chunks || m.windowed || !m.d_bt_entry) continue;
				batch.push_back(i);
				if (batch.size() == (size_t)BT_GROUP_MAX && !flush_bt()) { msg = "group backtrace launch failed"; abort_all(); return WHAMD_ERR_DEVICE; }
			}
			if (!flush_bt()) { msg = "group backtrace launch failed"; abort_all(); return WHAMD_ERR_DEVICE; }
		}
		for (size_t i : part.members) {
			Impl& m = *tables[i]->impl_;
			m.launches = table_launches[i];
			m.next_super = m.schedule.size();
			whamd_status_t st = WHAMD_OK;
			// The tail (backtrace, downloads) of every member goes onto the LEAD's stream, behind the group's last launch in the same hardware queue.  On the members'
			// own streams -- each waiting for the lead's event -- a 96-table step was bimodal: 67 ms or 95 ms, the device's forward pass 39 ms either way
			// (hardware queues that hold only a barrier are rescheduled late; more queues, GPU_MAX_HW_QUEUES=16, made every step 180 ms).  WHAMD_TAIL_OWN_STREAM=1
			// (debug library) restores the old placement.
			// ... except a member that walks back through the SEQUENTIAL kernel (several jobs, or too few units for chunks): milliseconds of one workgroup per table --
			// those run side by side on the members' own streams, behind the group's event, instead of one after the other on the lead's.
			const bool own = debug_env("WHAMD_TAIL_OWN_STREAM") != nullptr || (!walked[i] && !m.use_chunks && !m.windowed && m.stream != part.lead->stream);
			if (own) m.own_stream_used = true;
			if (own && m.stream != part.lead->stream && hipStreamWaitEvent(m.stream, part.lead->ev_group, 0) != hipSuccess) { msg = "hipStreamWaitEvent failed"; st = WHAMD_ERR_DEVICE; }
			if (st == WHAMD_OK) st = m.submit_tail(*problems[i], msg, own ? nullptr : part.lead->stream, walked[i] != 0, walked[i] != 0);
			if (st != WHAMD_OK) { abort_all(); return st; }
			m.enqueue_open = false;
			m.run_stream = m.stream;
		}
	}
	return WHAMD_OK;
}

// Tables in flight whose tails share a stream (a group: every member's tail is on the lead's stream, in order) finish in that order: ONE wait for the last of each
// stream, on the calling thread, and every table's own wait() returns at once.  (32 threads inside hipEventSynchronize at the same time woke up over 6 - 13 ms
// after the device had recorded the last event -- the events of one table 1.8 ms apart; one waiter wakes once.)
void DeviceTable::wait_last_of_each_stream(DeviceTable* const* tables, size_t n_tables) {
	std::vector<std::pair<hipStream_t, const Impl*>> last;
	for (size_t i = 0; i < n_tables; ++i) {
		const Impl& m = *tables[i]->impl_;
		if (!m.tail_elsewhere || !m.ev3) continue;
		bool found = false;
		for (auto& e : last) {
			if (e.first != m.tail_stream || e.second->device != m.device) continue;
			found = true;
			if (m.tail_seq > e.second->tail_seq) e.second = &m;
		}
		if (!found) last.emplace_back(m.tail_stream, &m);
	}
	for (const auto& e : last) {
		if (hipSetDevice(e.second->device) != hipSuccess || hipEventSynchronize(e.second->ev3) != hipSuccess) (void)hipGetLastError();   // (the table's own wait() reports it)
	}
}

// The event timings of the last collected solve (forward, backtrace, first to last event), filled into `st` once; a no-op when they have been
// read already or when a new solve has been submitted since (the events then belong to that one).
void DeviceTable::read_timing(whamd_solve_stats& st) {
	Impl& m = *impl_;
	if (!m.timing_pending) return;
	m.timing_pending = false;
	if (hipSetDevice(m.device) != hipSuccess) return;
	float f01 = 0, f12 = 0, f03 = 0;
	if (hipEventElapsedTime(&f01, m.ev0, m.ev1) != hipSuccess || hipEventElapsedTime(&f12, m.ev1, m.ev2) != hipSuccess || hipEventElapsedTime(&f03, m.ev0, m.ev3) != hipSuccess) {
		(void)hipGetLastError();
		return;
	}
	st.forward_ms = f01;
	st.backtrace_ms = f12;
	st.total_ms = f03;
}

whamd_status_t DeviceTable::wait(const Problem& p, Solution& s, whamd_solve_stats& st, std::string& msg) {
	Impl& m = *impl_;
	if (p.n_cols == 0) return WHAMD_OK;
	HIP_TRY(hipSetDevice(m.device));
	const uint64_t launches = m.launches;
	if (m.tail_elsewhere) HIP_TRY(hipEventSynchronize(m.ev3));   // (the last thing submit_tail recorded, on the stream the tail went to)
	else { HIP_TRY(hipStreamSynchronize(m.stream)); m.own_stream_used = false; }
	m.upload_pending = false;   // (the solve ran behind ev_upload: the uploads are there)
	const uint32_t n = p.n_cols;
	std::memcpy(s.path_index.data(), m.h_pinned, (size_t)n * 4);
	std::memcpy(s.path_trans.data(), m.h_pinned + n, (size_t)n * 4);
	s.optimal_score = m.h_pinned[2 * (size_t)n];
	if (m.device_superreads) {
		const uint32_t* q = m.h_pinned + m.super_off;
		const uint8_t* h = (const uint8_t*)(q + n);
		if (std::memchr(h, SUPERREAD_CONFLICT, (size_t)2 * n) == nullptr) {   // (a conflict: the host's loop runs and reports it, finish_solution)
			s.quality.resize(n);
			s.allele0.resize(n);
			s.allele1.resize(n);
			std::memcpy(s.quality.data(), q, (size_t)n * 4);
			std::memcpy(s.allele0.data(), h, n);
			std::memcpy(s.allele1.data(), h + n, n);
			s.superreads_done = true;
		}
	}
	for (size_t j = 1; j < m.jobs.size(); ++j) s.optimal_score += m.h_pinned[2 * (size_t)n + j];  // connected components solved as their own jobs
	// (the event timings are read when somebody asks -- read_timing(), from whamd_dptable_get_stats: three runtime calls per table, from every
	//  waiting thread at once, are a measurable part of collecting a 96-table step and most callers never look at them)
	st.forward_ms = st.backtrace_ms = st.total_ms = 0;
	m.timing_pending = true;
	st.forward_launches = launches;
	st.group_tables = m.group_tables;
	st.bt_chunks = st.bt_missed = st.bt_rewalked = 0;
	if (m.use_chunks) {
		const uint32_t* c = m.h_pinned + 2 * (size_t)n + m.jobs.size();   // (downloaded with the path)
		st.bt_chunks = (uint32_t)m.chunks.size();
		st.bt_missed = c[0];
		st.bt_rewalked = c[1];
		if (debug_env("WHAMD_BT_STATS")) fprintf(stderr, "[whamd backtrace] %zu chunks, %u guesses missed (%u of them only in the transmission value), %u units walked again (of %zu)\n", m.chunks.size(), c[0], c[2], c[1], m.units.size());
	}
	if (m.dp.dbg && m.use_slots) {
		std::vector<unsigned long long> d(m.splan.runs.size() * 48);
		HIP_TRY(hipMemcpy(d.data(), m.dp.dbg, d.size() * 8, hipMemcpyDeviceToHost));
		double a[6] = {0, 0, 0, 0, 0, 0};
		double percol[32] = {0};
		size_t cnt = 0;
		for (size_t i = 0; i < m.splan.runs.size(); ++i) {
			if (m.splan.runs[i].ncols != 22 || d[48 * i + 5] == 0) continue;   // full-length runs only
			for (int k = 0; k < 6; ++k) a[k] += (double)d[48 * i + k];
			for (int k = 0; k < 22; ++k) percol[k] += (double)d[48 * i + 8 + k];
			++cnt;
		}
		if (cnt) fprintf(stderr, "[whamd slot stamps] %zu runs, wave 0 of workgroup 0, shader cycles: prologue issue %.0f, loads landed %.0f, column loop %.0f (%.1f per column, %.1f columns), exit %.0f\n",
		                 cnt, a[0] / cnt, a[1] / cnt, a[2] / cnt, a[2] / std::max(a[4], 1.0), a[4] / cnt, a[3] / cnt);
		if (cnt) {
			fprintf(stderr, "[whamd slot stamps] cycles after the loop start at which column c had its cost added:");
			for (int k = 0; k < 22; ++k) fprintf(stderr, " %.0f", percol[k] / cnt);
			fprintf(stderr, "\n");
		}
	}
	if (m.dp.dbg && !m.plan.segments.empty()) {
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
		float f01 = 0;
		(void)hipEventElapsedTime(&f01, m.ev0, m.ev1);
		fprintf(stderr, "[whamd timing] segments %zu cols %llu | cycles/segment: prologue %.0f columns %.0f (%.0f per column) store %.0f | fwd %.3f ms, %.2f us per segment\n",
		        m.plan.segments.size(), cols, (double)a / m.plan.segments.size(), (double)b / m.plan.segments.size(),
		        (double)b / std::max<unsigned long long>(cols, 1), (double)c2 / m.plan.segments.size(), f01, f01 * 1e3 / m.plan.segments.size());
	}
	return WHAMD_OK;
}

void dptable_release_arena_cache() {
	std::vector<ArenaCache::Block> blocks;
	{
		std::lock_guard<std::mutex> lock(g_arena.mu);
		blocks.swap(g_arena.blocks);
	}
	for (const ArenaCache::Block& b : blocks) arena_free_block(b);
}

// whamd_release_caches: the kept backtrace arena AND the pinned upload staging area go back to the driver.
void dptable_release_caches() {
	dptable_release_arena_cache();
	devpool_release();
	misc_pool_release();
	std::lock_guard<std::mutex> lock(g_stage.mu);
	for (UploadStage::Area& a : g_stage.areas) {
		if (a.busy) continue;   // (an upload in flight on another thread keeps its area)
		if (a.parked && a.ev) (void)hipEventSynchronize(a.ev);   // (copies of a finished create may still be reading it)
		a.parked = false;
		if (a.base) (void)hipHostFree(a.base);
		a.base = nullptr;
		a.cap = 0;
	}
	g_stage.want = 0;
}

}  // namespace whamd
