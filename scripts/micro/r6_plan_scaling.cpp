// Round-6 probe: does the HOST part of a create (whamd_plan_summarize: flatten + plan, no device call) scale with the cores when every table has ONE thread of its own?
// No Python in the picture: std::threads call the C ABI directly; `pin` = 1 binds thread i to the i-th core of the calling process's CPU set (one thread per
// physical core: the first hardware thread of each), `pin` = 0 leaves placement to the scheduler.
// Build: g++ -O2 -std=c++17 -pthread -I include -o /tmp/r6ps scripts/micro/r6_plan_scaling.cpp -ldl ; run: WHAMD_PLAN_THREADS=1 /tmp/r6ps whatshap_amd/libwhatshap_amd.so
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <random>
#include <sched.h>
#include <sys/resource.h>
#include <thread>
#include <vector>
#include "whatshap_amd.h"

struct Table {
	std::vector<uint64_t> read_ptr; std::vector<int32_t> pos; std::vector<uint8_t> al; std::vector<uint32_t> q; std::vector<int32_t> sid;
	std::vector<uint32_t> recomb, positions, ind; std::vector<uint8_t> geno;
};
static Table make(uint32_t n, uint32_t cov, uint32_t seed) {   // reads of 30 consecutive columns, starts spread so that `cov` of them cover a column
	Table t; std::mt19937 rng(seed);
	const uint32_t len = 30;
	t.read_ptr.push_back(0);
	for (uint64_t i = 0;; ++i) {
		const uint64_t start = i * len / cov;
		if (start + 2 > n) break;
		for (uint32_t c = (uint32_t)start; c < std::min<uint64_t>(n, start + len); ++c) { t.pos.push_back((int32_t)(c * 10 + 1)); t.al.push_back(rng() & 1); t.q.push_back(5 + rng() % 30); }
		t.read_ptr.push_back(t.pos.size()); t.sid.push_back(7);
	}
	t.positions.resize(n); for (uint32_t c = 0; c < n; ++c) t.positions[c] = c * 10 + 1;
	t.recomb.assign(n, 10); t.geno.assign(n, 1); t.ind = {7};
	return t;
}
int main(int argc, char** argv) {
	void* lib = dlopen(argc > 1 ? argv[1] : "whatshap_amd/libwhatshap_amd.so", RTLD_NOW);
	if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 1; }
	auto summarize = (decltype(&whamd_plan_summarize))dlsym(lib, "whamd_plan_summarize");
	const int k = 96;
	std::vector<Table> tables;
	for (int i = 0; i < k; ++i) tables.push_back(make(50000, 15, 100 + i));
	cpu_set_t mine; sched_getaffinity(0, sizeof mine, &mine);
	std::vector<int> cores;   // first hardware thread of every core: on this box CPU c and c + 128 are siblings
	for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &mine) && c < 128) cores.push_back(c);
	auto one = [&](int i) {
		const Table& t = tables[i];
		whamd_readset_view rs{(uint32_t)t.sid.size(), t.read_ptr.data(), t.pos.data(), t.al.data(), t.q.data(), t.sid.data()};
		whamd_pedigree_view pv{1, t.ind.data(), 0, nullptr, (uint32_t)t.geno.size(), t.geno.data(), nullptr, nullptr};
		whamd_plan_summary out{};
		if (summarize(&rs, t.recomb.data(), t.recomb.size(), &pv, 0, t.positions.data(), t.positions.size(), "auto", &out) != WHAMD_OK) { fprintf(stderr, "summarize failed\n"); exit(1); }
	};
	one(0);
	for (int pin = 0; pin < 2; ++pin)
		for (int workers : {1, 16, 32, 64, 96}) {
			if (pin && workers > (int)cores.size()) continue;
			if (const char* only = getenv("R6_ONLY")) if (atoi(only) != workers || pin) continue;   // (one shape, scheduler's placement: for a phase-by-phase look with WHAMD_DEBUG_TIMING=1)
			double best = 1e30;
			const int reps = getenv("R6_REPS") ? atoi(getenv("R6_REPS")) : 3;
			for (int rep = 0; rep < reps; ++rep) {
				struct rusage u0; getrusage(RUSAGE_SELF, &u0);
				std::atomic<int> next{0};
				const auto t0 = std::chrono::steady_clock::now();
				std::vector<std::thread> th;
				for (int w = 0; w < workers; ++w) th.emplace_back([&, w] {
					if (pin) { cpu_set_t s; CPU_ZERO(&s); CPU_SET(cores[w], &s); sched_setaffinity(0, sizeof s, &s); }
					for (int i; (i = next.fetch_add(1)) < k;) one(i);
				});
				for (auto& x : th) x.join();
				const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
				best = std::min(best, ms);
				struct rusage u1; getrusage(RUSAGE_SELF, &u1);
				if (getenv("R6_REPS")) printf("    rep %d: %.1f ms, %ld minor page faults of the process, %.2f s user + %.2f s system time\n", rep, ms, u1.ru_minflt - u0.ru_minflt, (u1.ru_utime.tv_sec - u0.ru_utime.tv_sec) + 1e-6 * (u1.ru_utime.tv_usec - u0.ru_utime.tv_usec), (u1.ru_stime.tv_sec - u0.ru_stime.tv_sec) + 1e-6 * (u1.ru_stime.tv_usec - u0.ru_stime.tv_usec));
			}
			printf("%s %3d threads: %d tables in %7.1f ms = %6.0f tables/s, %5.1f thread-ms per table\n", pin ? "one per core  " : "scheduler's   ", workers, k, best, k / best * 1e3, best / k * std::min(workers, k));
			fflush(stdout);
		}
	return 0;
}
