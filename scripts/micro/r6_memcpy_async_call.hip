// Round-6 probe: how long the CALL hipMemcpyAsync(pinned -> device) takes on the calling thread, by size, alone and from 8 threads at once (own streams) --
// is a create's 28 MB copy "asynchronous" for the host thread, and do concurrent callers serialise?
//   hipcc --offload-arch=gfx950 -O3 -pthread -o /tmp/r6mc scripts/micro/r6_memcpy_async_call.hip && /tmp/r6mc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	for (size_t mb : {1, 8, 28, 110}) {
		const size_t bytes = mb << 20;
		for (int threads : {1, 8}) {
			std::vector<double> call_us(threads, 0.0), total_us(threads, 0.0);
			std::vector<std::thread> pool;
			const double t_all0 = now_us();
			for (int t = 0; t < threads; ++t) pool.emplace_back([&, t] {
				void *h, *d; hipStream_t s;
				(void)hipHostMalloc(&h, bytes, hipHostMallocPortable); (void)hipMalloc(&d, bytes); (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
				(void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s);   // warm
				const int reps = 20;
				double call = 0, total = 0;
				for (int r = 0; r < reps; ++r) {
					const double a = now_us();
					(void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
					const double b = now_us();
					(void)hipStreamSynchronize(s);
					const double c = now_us();
					call += b - a; total += c - a;
				}
				call_us[t] = call / reps; total_us[t] = total / reps;
				(void)hipFree(d); (void)hipHostFree(h); (void)hipStreamDestroy(s);
			});
			for (auto& th : pool) th.join();
			double call = 0, total = 0;
			for (int t = 0; t < threads; ++t) { call += call_us[t] / threads; total += total_us[t] / threads; }
			printf("%4zu MB, %d thread(s): the call returns after %8.1f us, the copy is done after %8.1f us (%.1f GB/s per thread, %.1f aggregate)\n", mb, threads, call, total,
			       bytes / total / 1e3, threads * bytes / total / 1e3);
			(void)t_all0;
		}
	}
	return 0;
}
