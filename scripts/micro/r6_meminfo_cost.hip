// Round-6 probe: what hipMemGetInfo costs per call, alone and from 16 threads at once (a create asks once, a release asked once per arena).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	void* big; (void)hipMalloc(&big, (size_t)8 << 30);
	for (int threads : {1, 16}) {
		std::vector<double> per(threads);
		std::vector<std::thread> pool;
		for (int t = 0; t < threads; ++t) pool.emplace_back([&, t] {
			size_t f, tot; (void)hipSetDevice(0); (void)hipMemGetInfo(&f, &tot);
			const double a = now_us();
			for (int i = 0; i < 200; ++i) (void)hipMemGetInfo(&f, &tot);
			per[t] = (now_us() - a) / 200;
		});
		for (auto& th : pool) th.join();
		double s = 0; for (double v : per) s += v / threads;
		printf("hipMemGetInfo: %.1f us per call with %d thread(s) calling at once\n", s, threads);
	}
	return 0;
}
