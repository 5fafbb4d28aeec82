// Round-6 probe: how long hipEventSynchronize takes on an event recorded LONG AGO behind a host -> device copy and a small kernel -- on a normal stream and on a stream of the
// highest priority -- when the calling thread has done other device work (a solve on another stream) in between.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(int* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
int main() {
	int least = 0, greatest = 0;
	(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
	hipStream_t normal, high, work;
	(void)hipStreamCreateWithFlags(&normal, hipStreamNonBlocking);
	(void)hipStreamCreateWithPriority(&high, hipStreamNonBlocking, greatest);
	(void)hipStreamCreateWithFlags(&work, hipStreamNonBlocking);
	const size_t bytes = (size_t)32 << 20;
	void *h, *d; int* dw;
	(void)hipHostMalloc(&h, bytes, hipHostMallocPortable); (void)hipMalloc(&d, bytes); (void)hipMalloc(&dw, 4096 * 4); (void)hipMemset(dw, 0, 4096 * 4);
	hipEvent_t e;
	(void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
	for (int which = 0; which < 2; ++which) {
		hipStream_t s = which ? high : normal;
		for (int rep = 0; rep < 4; ++rep) {
			(void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
			hipLaunchKernelGGL(touch, dim3(64), dim3(64), 0, s, dw);
			(void)hipEventRecord(e, s);
			for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(touch, dim3(8), dim3(64), 0, work, dw);   // "a solve" on another stream
			(void)hipStreamSynchronize(work);
			std::this_thread::sleep_for(std::chrono::milliseconds(5));
			const double t0 = now_ms();
			(void)hipEventSynchronize(e);
			printf("%s stream, rep %d: hipEventSynchronize on an event recorded long ago took %.3f ms (priority range %d..%d)\n", which ? "high-priority" : "normal", rep, now_ms() - t0, least, greatest);
		}
	}
	return 0;
}
