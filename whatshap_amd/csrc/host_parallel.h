// host_parallel.h -- the two helpers of the host create path (problem.cpp, slot_plan.cpp, dp_device.hip): ranges of independent items
// on a few host threads, and vectors that are not zero-filled when they are sized (the range workers touch their own part first, so
// the page faults of a 30 MB array are spread over the threads as well).
#pragma once
#include <algorithm>
#include <cstdint>
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
#include <vector>

namespace whamd {

// Host threads for `n_items` independent items, at least `grain` items per thread: min(hardware threads, 32), or what
// WHAMD_PLAN_THREADS says (bench.py reports the create path at 8 threads and at the default).
// (per calling thread: a create that runs beside many others -- blocks.solve_blocks builds a window's tables on a pool of host threads -- is told
// to keep to a few threads of its own, option "host_threads" of whamd_dptable_create_with_options; 0 = no override)
inline uint32_t& host_threads_override() {
	static thread_local uint32_t value = 0;
	return value;
}
inline uint32_t host_threads(uint64_t n_items, uint64_t grain) {
	uint32_t n_threads = std::min<uint32_t>(std::max(1u, std::thread::hardware_concurrency()), 32u);
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

// fn(begin, end, t) for n_threads contiguous ranges of [0, n); the calling thread takes the last range.
// Exception-safe: the guard that joins the workers and gives the caller its affinity mask back exists BEFORE the first worker does; a
// worker that cannot be started (std::system_error: EAGAIN under a pids limit) leaves its range and the following ones to the caller;
// an exception inside a worker's fn is carried to the caller and rethrown after the join (the C ABI turns it into a status).
template <class F>
inline void parallel_ranges(uint64_t n, uint32_t n_threads, F&& fn) {
	if (n_threads <= 1) {
		fn((uint64_t)0, n, 0u);
		return;
	}
	const cpu_set_t* node_cpus = caller_node_cpus();
	// (the calling thread takes the last range: it joins the workers on their node for the duration)
	cpu_set_t caller_mask;
	const bool rebind = node_cpus && pthread_getaffinity_np(pthread_self(), sizeof caller_mask, &caller_mask) == 0 &&
	                    pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), node_cpus) == 0;
	std::vector<std::thread> workers;
	std::vector<std::exception_ptr> failed(n_threads);
	struct Finish {   // on every way out: join, give the caller its mask back
		std::vector<std::thread>& workers;
		const bool rebind;
		const cpu_set_t& mask;
		~Finish() {
			for (std::thread& w : workers) if (w.joinable()) w.join();
			if (rebind) (void)pthread_setaffinity_np(pthread_self(), sizeof mask, &mask);
		}
	};
	{
		Finish finish{workers, rebind, caller_mask};
		workers.reserve(n_threads - 1);
		uint32_t started = 0;
		for (; started + 1 < n_threads; ++started) {
			const uint32_t t = started;
			const uint32_t budget = host_threads_override();   // thread_local: a worker that sizes a nested range keeps to the caller's budget
			try {
				workers.emplace_back([&fn, &failed, n, n_threads, t, node_cpus, budget]() {
					host_threads_override() = budget;
					if (node_cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), node_cpus);
					try {
						fn(n * t / n_threads, n * (t + 1) / n_threads, t);
					} catch (...) {
						failed[t] = std::current_exception();
					}
				});
			} catch (const std::system_error&) {
				break;   // no more threads to be had: the caller runs what is left
			}
		}
		for (uint32_t t = started; t < n_threads; ++t) fn(n * t / n_threads, t + 1 == n_threads ? n : n * (t + 1) / n_threads, t);
	}
	for (const std::exception_ptr& e : failed)
		if (e) std::rethrow_exception(e);
}

// Allocator of the create path's large arrays: the value-less construct() default-initialises (resize() of a vector of trivial
// elements does not write them), and blocks of 4 MB and more are 2 MB-aligned and advised as transparent huge pages -- a fresh
// 50 MB array is then 25 page faults instead of 12 800 (the range workers fault their own parts in; with 4 KB pages the faults
// of 16 threads serialise in the kernel and were most of the flatten / plan time).
template <class T>
struct NoInitAlloc {
	using value_type = T;
	static constexpr size_t HUGE_FROM = (size_t)4 << 20, HUGE_PAGE = (size_t)2 << 20;
	NoInitAlloc() = default;
	template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
	template <class U> struct rebind { using other = NoInitAlloc<U>; };
	T* allocate(size_t n) {
		const size_t bytes = n * sizeof(T);
		if (bytes >= HUGE_FROM) {
			const size_t rounded = (bytes + HUGE_PAGE - 1) / HUGE_PAGE * HUGE_PAGE;
			void* ptr = std::aligned_alloc(HUGE_PAGE, rounded);
			if (!ptr) throw std::bad_alloc();
			static const bool advise = getenv("WHAMD_NO_HUGEPAGES") == nullptr;
			if (advise) (void)madvise(ptr, rounded, MADV_HUGEPAGE);
			return static_cast<T*>(ptr);
		}
		return static_cast<T*>(::operator new(bytes));
	}
	void deallocate(T* ptr, size_t n) noexcept {
		if (n * sizeof(T) >= HUGE_FROM) std::free(ptr);
		else ::operator delete(ptr);
	}
	template <class U> void construct(U* ptr) { ::new ((void*)ptr) U; }
	template <class U, class... A> void construct(U* ptr, A&&... args) { ::new ((void*)ptr) U(std::forward<A>(args)...); }
	template <class U> bool operator==(const NoInitAlloc<U>&) const { return true; }
	template <class U> bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <class T>
using RawVec = std::vector<T, NoInitAlloc<T>>;

}  // namespace whamd
