// host_memory.cpp -- the host side's large blocks are KEPT between tables (the device side has its pools: device_pool.h).
//
// A table's create path sizes ~100 arrays of 64 KB .. 50 MB (columns, entries, deltas, terms, plan rows, descriptors, solution) and its destroy
// frees them.  glibc serves such sizes with one mmap each and returns them with munmap: every create faults its pages in again, every destroy
// tears them down, and both serialise on the process's address-space lock -- 0.7 ms of munmap per coverage-15 table in whamd_dptable_destroy
// (68 ms for the 96 tables of one step), 5.8 ms for configs[2]'s 200 MB, and the reason sixteen concurrent creates did not scale to sixty-four
// (measured: profiles/r06/, MALLOC_MMAP_THRESHOLD_ experiment).  This file replaces the global allocation functions FOR THIS SHARED OBJECT ONLY
// (the link's version script keeps every symbol but whamd_* local and binds references inside the library: Python, libstdc++ and HIP keep the
// process's own malloc): requests below 64 KB go to malloc unchanged; larger ones are rounded to a size class (eight per octave: at most 12.5 %
// slack), served from the idle blocks of that class, and returned to them -- up to WHAMD_HOST_POOL_MB (default 16384 -- 96 coverage-15 tables hold 4.3 GB; 0 switches the pool off), the
// rest goes back to the system.  Blocks of 2 MB and more are 2 MB-aligned and advised as transparent huge pages (a fresh 50 MB array is 25 page
// faults instead of 12 800).  Every block comes from posix_memalign, so a pointer that leaves through somebody else's free() is still valid C;
// a pointer this library frees that it did not allocate (a std::string grown inside libstdc++.so) is recognised by the block table and handed to
// free().  whamd_release_caches() empties the pool.
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <malloc.h>
#include <sched.h>
#include <mutex>
#include <new>
#include <sys/mman.h>
#include <unordered_map>
#include <vector>

#include "host_parallel.h"

namespace whamd {

namespace {

constexpr size_t POOL_FROM = (size_t)64 << 10, HUGE_PAGE = (size_t)2 << 20, SMALL_PAGE = 4096;

// The pool's lock is held for a hash look-up and a vector push or pop (~100 ns), ~200 times per table.  As a std::mutex it made a convoy of sixty-four concurrent
// creates: a waiter SLEEPS, and every hand-over then costs a futex wake (21 us of waiting per acquisition measured, 2 - 6 s in total for 480 tables on 32 - 64
// threads: threads asleep half of the time, scripts/micro/r6_plan_scaling.cpp).  Spinning for a critical section this short costs its length.
struct SpinLock {
	std::atomic<bool> held{false};
	void lock() {
		for (uint32_t spins = 0;; ++spins) {
			if (!held.load(std::memory_order_relaxed) && !held.exchange(true, std::memory_order_acquire)) return;
			if (spins < 4096) __builtin_ia32_pause();
			else sched_yield();
		}
	}
	void unlock() { held.store(false, std::memory_order_release); }
};

struct HostPool {
	SpinLock mu;
	std::unordered_map<size_t, std::vector<void*>> idle;   // class size -> blocks
	std::unordered_map<void*, size_t> size_of;            // every block handed out or idle -> its class size
	size_t idle_bytes = 0, keep = 0;
	bool advise = true;
	HostPool() {
		const char* e = getenv("WHAMD_HOST_POOL_MB");
		keep = (size_t)(e ? std::max(0, atoi(e)) : 16384) << 20;
		advise = getenv("WHAMD_NO_HUGEPAGES") == nullptr;
	}
};
HostPool& pool() {
	static HostPool* p = new HostPool();   // (never destroyed: vectors are freed during static destruction too)
	return *p;
}
thread_local bool g_inside = false;
size_t class_of(size_t bytes) {
	size_t top = (size_t)1 << (63 - __builtin_clzll((unsigned long long)bytes));   // largest power of two <= bytes
	size_t granule = std::max(SMALL_PAGE, top >> 3);
	size_t rounded = (bytes + granule - 1) / granule * granule;
	if (rounded >= HUGE_PAGE) rounded = (rounded + HUGE_PAGE - 1) / HUGE_PAGE * HUGE_PAGE;
	return rounded;
}

}  // namespace

bool host_pool_enabled() { return pool().keep != 0; }

void* host_pool_take(size_t bytes) {
	HostPool& p = pool();
	const size_t size = class_of(std::max(bytes, POOL_FROM));
	struct Inside { Inside() { g_inside = true; } ~Inside() { g_inside = false; } } inside;
	{
		std::lock_guard<SpinLock> lock(p.mu);
		auto it = p.idle.find(size);
		if (it != p.idle.end() && !it->second.empty()) {
			void* ptr = it->second.back();
			it->second.pop_back();
			p.idle_bytes -= size;
			return ptr;
		}
	}
	void* ptr = nullptr;
	if (posix_memalign(&ptr, size >= HUGE_PAGE ? HUGE_PAGE : SMALL_PAGE, size) != 0 || !ptr) return nullptr;
	if (size >= HUGE_PAGE && p.advise) (void)madvise(ptr, size, MADV_HUGEPAGE);
	std::lock_guard<SpinLock> lock(p.mu);
	p.size_of[ptr] = size;
	return ptr;
}

// true: the block was one of the pool's (kept or given back to the system); false: not ours, the caller frees it
bool host_pool_give(void* ptr) {
	HostPool& p = pool();
	struct Inside { Inside() { g_inside = true; } ~Inside() { g_inside = false; } } inside;
	bool release = false;
	{
		std::lock_guard<SpinLock> lock(p.mu);
		const auto it = p.size_of.find(ptr);
		if (it == p.size_of.end()) return false;
		const size_t size = it->second;
		// (a block that left through somebody else's free() and came back as a smaller allocation of theirs: never trust the table alone)
		if (malloc_usable_size(ptr) < size || p.idle_bytes + size > p.keep) {
			p.size_of.erase(it);
			release = true;
		} else {
			p.idle[size].push_back(ptr);
			p.idle_bytes += size;
		}
	}
	if (release) std::free(ptr);
	return true;
}

void host_pool_release() {
	HostPool& p = pool();
	struct Inside { Inside() { g_inside = true; } ~Inside() { g_inside = false; } } inside;
	std::vector<void*> drop;
	{
		std::lock_guard<SpinLock> lock(p.mu);
		for (auto& kv : p.idle) {
			for (void* ptr : kv.second) { p.size_of.erase(ptr); drop.push_back(ptr); }
			kv.second.clear();
		}
		p.idle_bytes = 0;
	}
	for (void* ptr : drop) std::free(ptr);
}

size_t host_pool_idle_bytes() {
	HostPool& p = pool();
	std::lock_guard<SpinLock> lock(p.mu);
	return p.idle_bytes;
}

}  // namespace whamd

// ---- the replacement allocation functions (local to this shared object: csrc/exports.map)
static inline void* whamd_allocate(std::size_t size) {
	if (size >= whamd::POOL_FROM && !whamd::g_inside && whamd::host_pool_enabled()) return whamd::host_pool_take(size);
	return std::malloc(size ? size : 1);
}
static inline void whamd_deallocate(void* ptr) noexcept {
	if (!ptr) return;
	// pool blocks are page-aligned; the test spares nearly every small free the look-up (a malloc'd chunk is page-aligned once in 256)
	if ((reinterpret_cast<std::uintptr_t>(ptr) & (whamd::SMALL_PAGE - 1)) == 0 && !whamd::g_inside && whamd::host_pool_give(ptr)) return;
	std::free(ptr);
}
void* operator new(std::size_t size) {
	void* ptr = whamd_allocate(size);
	if (!ptr) throw std::bad_alloc();
	return ptr;
}
void* operator new[](std::size_t size) {
	void* ptr = whamd_allocate(size);
	if (!ptr) throw std::bad_alloc();
	return ptr;
}
void* operator new(std::size_t size, const std::nothrow_t&) noexcept { return whamd_allocate(size); }
void* operator new[](std::size_t size, const std::nothrow_t&) noexcept { return whamd_allocate(size); }
void operator delete(void* ptr) noexcept { whamd_deallocate(ptr); }
void operator delete[](void* ptr) noexcept { whamd_deallocate(ptr); }
void operator delete(void* ptr, std::size_t) noexcept { whamd_deallocate(ptr); }
void operator delete[](void* ptr, std::size_t) noexcept { whamd_deallocate(ptr); }
void operator delete(void* ptr, const std::nothrow_t&) noexcept { whamd_deallocate(ptr); }
void operator delete[](void* ptr, const std::nothrow_t&) noexcept { whamd_deallocate(ptr); }
