// host_parallel.h -- the two helpers of the host create path (problem.cpp, slot_plan.cpp, dp_device.hip): ranges of independent items
// on a few host threads, and vectors that are not zero-filled when they are sized (the range workers touch their own part first, so
// the page faults of a 30 MB array are spread over the threads as well).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <new>
#include <sys/mman.h>
#include <thread>
#include <vector>

namespace whamd {

// Host threads for `n_items` independent items, at least `grain` items per thread: min(hardware threads, 32), or what
// WHAMD_PLAN_THREADS says (bench.py reports the create path at 8 threads and at the default).
inline uint32_t host_threads(uint64_t n_items, uint64_t grain) {
	uint32_t n_threads = std::min<uint32_t>(std::max(1u, std::thread::hardware_concurrency()), 32u);
	if (const char* e = getenv("WHAMD_PLAN_THREADS")) n_threads = (uint32_t)std::max(1, atoi(e));
	return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_threads, n_items / std::max<uint64_t>(grain, 1) + 1));
}

// fn(begin, end, t) for n_threads contiguous ranges of [0, n); the calling thread takes the last range.
template <class F>
inline void parallel_ranges(uint64_t n, uint32_t n_threads, F&& fn) {
	if (n_threads <= 1) {
		fn((uint64_t)0, n, 0u);
		return;
	}
	std::vector<std::thread> workers;
	workers.reserve(n_threads - 1);
	for (uint32_t t = 0; t + 1 < n_threads; ++t) workers.emplace_back([&fn, n, n_threads, t]() { fn(n * t / n_threads, n * (t + 1) / n_threads, t); });
	fn(n * (n_threads - 1) / n_threads, n, n_threads - 1);
	for (std::thread& w : workers) w.join();
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
