#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <sys/resource.h>
int main() {
	for (int workers : {8, 16, 32, 64, 128}) {
		struct rusage u0; getrusage(RUSAGE_SELF, &u0);
		const auto t0 = std::chrono::steady_clock::now();
		std::vector<std::thread> th;
		for (int w = 0; w < workers; ++w) th.emplace_back([] { volatile double x = 1; const auto a = std::chrono::steady_clock::now(); while (std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() < 0.2) for (int i = 0; i < 1000; ++i) x = x * 1.0000001 + 1e-9; });
		for (auto& x : th) x.join();
		struct rusage u1; getrusage(RUSAGE_SELF, &u1);
		const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		const double cpu = (u1.ru_utime.tv_sec - u0.ru_utime.tv_sec) + 1e-6 * (u1.ru_utime.tv_usec - u0.ru_utime.tv_usec);
		printf("%3d spinning threads for 0.2 s: wall %.3f s, CPU time %.2f s = %.1f CPUs\n", workers, wall, cpu, cpu / wall);
	}
}
