// host_parallel.h -- the two helpers of the host create path (problem.cpp, slot_plan.cpp, dp_device.hip): ranges of independent items
// on a few host threads, and vectors that are not zero-filled when they are sized (the range workers touch their own part first, so
// the page faults of a 30 MB array are spread over the threads as well).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <system_error>
#include <thread>
#include <string>
#include <vector>
#include <sys/resource.h>

namespace whamd {

// Host threads for `n_items` independent items, at least `grain` items per thread: min(hardware threads, 32), or what
// WHAMD_PLAN_THREADS says (bench.py reports the create path at 8 threads and at the default).
// (per calling thread: a create that runs beside many others -- blocks.solve_blocks builds a window's tables on a pool of host threads -- is told
// to keep to a few threads of its own, option "host_threads" of whamd_dptable_create_with_options; 0 = no override)
// Minor page faults of the calling thread so far (WHAMD_DEBUG_TIMING laps: a create that faults its memory in again is bound by the kernel's address-space locks).
inline long thread_minor_faults() {
	struct rusage u;
	return getrusage(RUSAGE_THREAD, &u) == 0 ? u.ru_minflt : 0;
}

inline uint32_t& host_threads_override() {
	static thread_local uint32_t value = 0;
	return value;
}
// CPUs this process may keep busy: its affinity mask (a rank bound to the CPU slice of its GPU -- whatshap_amd.blocks.bind_rank_to_device_cpus -- sizes its workers by
// the slice, not by the machine), capped by WHAMD_HOST_CPUS; read once.  NOT capped by the control group's CPU quota (cgroup cpu.max: the MI355X boxes of this project
// give a job all 256 hardware threads in its mask and 16 CPUs of time per 100 ms -- whatshap_amd.blocks.cpu_quota reads it, scripts/micro/r6_cpu_quota_probe.cpp
// shows it): the quota limits CPU time per period, not threads --
// one create is a burst of a few thread-milliseconds and is fastest on all the threads it can use (configs[2]: 14.0 ms on 32 threads, 16.5 on 16); it is the SUSTAINED
// work of many creates that has to stay under the quota, and that is decided where the tables are queued (whatshap_amd.blocks.solve_blocks, bench.py's host shapes).
inline uint32_t usable_cpus() {
	static const uint32_t n = [] {
		cpu_set_t allowed;
		CPU_ZERO(&allowed);
		uint32_t cpus = std::max(1u, std::thread::hardware_concurrency());
		if (sched_getaffinity(0, sizeof allowed, &allowed) == 0 && CPU_COUNT(&allowed) > 0) cpus = (uint32_t)CPU_COUNT(&allowed);
		if (const char* e = getenv("WHAMD_HOST_CPUS")) if (atoi(e) > 0) cpus = std::min<uint32_t>(cpus, (uint32_t)atoi(e));
		return std::max(1u, cpus);
	}();
	return n;
}
inline uint32_t host_threads(uint64_t n_items, uint64_t grain) {
	uint32_t n_threads = std::min<uint32_t>(usable_cpus(), 32u);
	if (const char* e = getenv("WHAMD_PLAN_THREADS")) n_threads = (uint32_t)std::max(1, atoi(e));
	if (host_threads_override()) n_threads = host_threads_override();
	return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_threads, n_items / std::max<uint64_t>(grain, 1) + 1));
}

// The CPUs of the NUMA node the calling thread runs on (intersected with what the process may use), or an empty set.  Workers are
// bound to that set: on the two-socket host of the MI355X box unbound workers landed on both sockets, away from the arrays the caller
// allocated -- the flattener's pass took 16.9 ms on 8 threads, 5.3 ms with the process held to one node (measured with taskset).
inline const cpu_set_t* caller_node_cpus() {
	struct Nodes {
		std::vector<cpu_set_t> sets;      // per node
		std::vector<int> node_of_cpu;
		Nodes() {
			if (getenv("WHAMD_NO_AFFINITY")) return;
			cpu_set_t allowed;
			CPU_ZERO(&allowed);
			if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
			node_of_cpu.assign(CPU_SETSIZE, -1);
			for (int node = 0; node < 64; ++node) {
				char path[96];
				snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
				FILE* f = fopen(path, "r");
				if (!f) break;
				char text[4096];
				const bool got = fgets(text, sizeof text, f) != nullptr;
				fclose(f);
				cpu_set_t set;
				CPU_ZERO(&set);
				if (got) {
					for (char* tok = strtok(text, ",\n"); tok; tok = strtok(nullptr, ",\n")) {   // "0-63,128-191"
						int lo = 0, hi = 0;
						const int fields = sscanf(tok, "%d-%d", &lo, &hi);
						if (fields == 1) hi = lo;
						if (fields < 1) continue;
						for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c)
							if (c >= 0 && CPU_ISSET(c, &allowed)) { CPU_SET(c, &set); node_of_cpu[c] = node; }
					}
				}
				sets.push_back(set);
			}
			if (sets.size() < 2) sets.clear();   // one node: nothing to choose
		}
	};
	static const Nodes nodes;   // (strtok above runs once, under the static's initialisation lock)
	if (nodes.sets.empty()) return nullptr;
	// ONE node per process, the one the first caller ran on: the arrays of later tables are then touched where the earlier ones were
	static const int chosen = [] {
		const int cpu = sched_getcpu();
		return cpu >= 0 && cpu < (int)nodes.node_of_cpu.size() ? nodes.node_of_cpu[cpu] : -1;
	}();
	if (chosen < 0) return nullptr;
	const cpu_set_t* set = &nodes.sets[chosen];
	return CPU_COUNT(set) > 0 ? set : nullptr;
}

// ---- a persistent pool of host workers.  parallel_ranges used to start and join n - 1 std::threads per call: ~40 us each, ten calls per create --
// a millisecond of the caller's time per call at 32 threads, and a clone / mmap / munmap per worker of CPU time.  The workers now exist once per
// process (started on first use, bound to the caller's NUMA node like the per-call threads were) and sleep on a condition variable between calls.
// A call publishes ONE job (ranges are handed out by an atomic counter) and n - 1 tickets for it; the caller takes ranges itself until none is
// left -- so a call never waits for a worker to wake up, nested calls cannot deadlock, and a pool that is busy with other tables' creates simply
// leaves more ranges to the caller.  WHAMD_NO_WORKER_POOL=1: the previous thread-per-call behaviour.
struct RangeJob {
	void (*run)(void* ctx, uint64_t begin, uint64_t end, uint32_t t) = nullptr;
	void* ctx = nullptr;
	uint64_t n = 0;
	uint32_t n_threads = 0, budget = 0;
	std::atomic<uint32_t> next{0}, done{0};
	std::mutex mu;                      // guards `failed` and the completion wait
	std::condition_variable finished;
	std::exception_ptr failed;
	// one range; returns false when none is left
	bool take_one() {
		const uint32_t t = next.fetch_add(1, std::memory_order_relaxed);
		if (t >= n_threads) return false;
		try {
			run(ctx, n * t / n_threads, t + 1 == n_threads ? n : n * (t + 1) / n_threads, t);
		} catch (...) {
			std::lock_guard<std::mutex> lock(mu);
			if (!failed) failed = std::current_exception();
		}
		if (done.fetch_add(1, std::memory_order_acq_rel) + 1 == n_threads) {
			std::lock_guard<std::mutex> lock(mu);
			finished.notify_all();
		}
		return true;
	}
};

class WorkerPool {
public:
	static WorkerPool& instance() {
		static WorkerPool* pool = new WorkerPool();   // (never destroyed: workers may be asleep in it when the process exits)
		return *pool;
	}
	// tickets for `extra` more workers on `job`
	void offer(const std::shared_ptr<RangeJob>& job, uint32_t extra) {
		{
			std::lock_guard<std::mutex> lock(mu_);
			grow(extra);
			for (uint32_t i = 0; i < extra; ++i) tickets_.push_back(job);
		}
		for (uint32_t i = 0; i < extra; ++i) wake_.notify_one();   // (one sleeper per ticket: waking every worker of a 256-thread host for two tickets is a herd)
	}
private:
	std::mutex mu_;
	std::condition_variable wake_;
	std::deque<std::shared_ptr<RangeJob>> tickets_;
	uint32_t n_workers_ = 0, idle_ = 0;
	void grow(uint32_t wanted) {   // mu_ held: at most as many workers as CPUs this process may use
		const uint32_t cap = std::max(1u, usable_cpus());   // (many tables are created at once, each with a few workers: blocks.solve_blocks)
		const uint32_t pending = (uint32_t)tickets_.size() + wanted;   // tickets waiting for a worker once `wanted` more are queued
		while (n_workers_ < cap && idle_ < pending) {
			try {
				std::thread([this] { loop(); }).detach();
			} catch (const std::system_error&) {
				break;   // no more threads to be had: the callers run what is left themselves
			}
			++n_workers_;
			++idle_;     // (counts as idle until it picks its first ticket)
		}
	}
	void loop() {
		if (const cpu_set_t* node_cpus = caller_node_cpus()) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), node_cpus);
		std::unique_lock<std::mutex> lock(mu_);
		for (;;) {
			while (tickets_.empty()) wake_.wait(lock);
			std::shared_ptr<RangeJob> job = std::move(tickets_.front());
			tickets_.pop_front();
			--idle_;
			lock.unlock();
			const uint32_t saved = host_threads_override();
			host_threads_override() = job->budget;   // thread_local: a worker that sizes a nested range keeps to the caller's budget
			while (job->take_one()) {}
			host_threads_override() = saved;
			job.reset();
			lock.lock();
			++idle_;
		}
	}
};

// fn(begin, end, t) for n_threads contiguous ranges of [0, n); range t is executed exactly once, by the caller or by a pool worker.
// Exception-safe: an exception inside fn is carried to the caller and rethrown after every range has finished (the C ABI turns it into a
// status); the caller gets its affinity mask back on every way out.
template <class F>
inline void parallel_ranges(uint64_t n, uint32_t n_threads, F&& fn) {
	if (n_threads <= 1) {
		fn((uint64_t)0, n, 0u);
		return;
	}
	static const bool no_pool = getenv("WHAMD_NO_WORKER_POOL") != nullptr;
	const cpu_set_t* node_cpus = caller_node_cpus();
	// (the calling thread joins the workers on their node for the duration)
	cpu_set_t caller_mask;
	const bool rebind = node_cpus && pthread_getaffinity_np(pthread_self(), sizeof caller_mask, &caller_mask) == 0 &&
	                    pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), node_cpus) == 0;
	struct Restore {
		const bool rebind;
		const cpu_set_t& mask;
		~Restore() { if (rebind) (void)pthread_setaffinity_np(pthread_self(), sizeof mask, &mask); }
	} restore{rebind, caller_mask};
	using Fn = typename std::remove_reference<F>::type;
	auto job = std::make_shared<RangeJob>();
	job->run = [](void* ctx, uint64_t b, uint64_t e, uint32_t t) { (*static_cast<Fn*>(ctx))(b, e, t); };
	job->ctx = const_cast<void*>(static_cast<const void*>(&fn));
	job->n = n;
	job->n_threads = n_threads;
	job->budget = host_threads_override();
	std::vector<std::thread> own;
	if (no_pool) {
		for (uint32_t i = 0; i + 1 < n_threads; ++i) {
			try {
				own.emplace_back([job, node_cpus] {
					host_threads_override() = job->budget;
					if (node_cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), node_cpus);
					while (job->take_one()) {}
				});
			} catch (const std::system_error&) {
				break;
			}
		}
	} else {
		WorkerPool::instance().offer(job, n_threads - 1);
	}
	while (job->take_one()) {}
	for (std::thread& w : own) w.join();
	if (job->done.load(std::memory_order_acquire) != n_threads) {   // ranges still running on workers: `fn` and its captures must outlive them
		std::unique_lock<std::mutex> lock(job->mu);
		job->finished.wait(lock, [&] { return job->done.load(std::memory_order_acquire) == n_threads; });
	}
	if (job->failed) std::rethrow_exception(job->failed);
}

// Allocator of the create path's large arrays: the value-less construct() default-initialises (resize() of a vector of trivial
// elements does not write them), and blocks of 4 MB and more are 2 MB-aligned and advised as transparent huge pages -- a fresh
// 50 MB array is then 25 page faults instead of 12 800 (the range workers fault their own parts in; with 4 KB pages the faults
// of 16 threads serialise in the kernel and were most of the flatten / plan time).
// The blocks themselves come from the library's own allocation functions (host_memory.cpp): requests of 64 KB and more are served from blocks
// kept between tables, 2 MB-aligned and advised as huge pages from 2 MB on.
void* host_pool_take(size_t bytes);
bool host_pool_give(void* ptr);
void host_pool_release();      // everything idle goes back to the system (whamd_release_caches)
size_t host_pool_idle_bytes();
bool host_pool_enabled();

template <class T>
struct NoInitAlloc {
	using value_type = T;
	NoInitAlloc() = default;
	template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
	template <class U> struct rebind { using other = NoInitAlloc<U>; };
	T* allocate(size_t n) { return static_cast<T*>(::operator new(n * sizeof(T))); }
	void deallocate(T* ptr, size_t) noexcept { ::operator delete(ptr); }
	template <class U> void construct(U* ptr) { ::new ((void*)ptr) U; }
	template <class U, class... A> void construct(U* ptr, A&&... args) { ::new ((void*)ptr) U(std::forward<A>(args)...); }
	template <class U> bool operator==(const NoInitAlloc<U>&) const { return true; }
	template <class U> bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <class T>
using RawVec = std::vector<T, NoInitAlloc<T>>;

}  // namespace whamd
