// Microbenchmark 2: what in a kernel lengthens the launch-to-launch boundary (stores at the end, loads at the start,
// large kernel arguments)?  Reported: microseconds per launch minus the same kernel's pure spin time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { unsigned a[90]; };
template <bool LOAD, bool STORE, bool ARGS>
__global__ __launch_bounds__(1024) void k(unsigned* p, unsigned* q, int spin, Big big) {
	extern __shared__ unsigned sm[];
	unsigned v = threadIdx.x;
	if (LOAD) { for (int i = 0; i < 4; ++i) v += q[(blockIdx.x * 4 + i) * 1024 + threadIdx.x]; }
	if (ARGS) v += big.a[threadIdx.x % 90];
	for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;
	if (STORE) { for (int i = 0; i < 4; ++i) p[(blockIdx.x * 4 + i) * 1024 + threadIdx.x] = v + i; }
	if (v == 0x12345678u) p[blockIdx.x] = v + sm[0];
}
template <bool LOAD, bool STORE, bool ARGS>
float run(hipStream_t s, unsigned* a, unsigned* b, int spin, hipEvent_t e0, hipEvent_t e1) {
	Big big{};
	const int N = 1000;
	float ms = 0;
	for (int rep = 0; rep < 2; ++rep) {
		(void)hipEventRecord(e0, s);
		for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k<LOAD, STORE, ARGS>), dim3(256), dim3(1024), 112 * 1024, s, (i & 1) ? a : b, (i & 1) ? b : a, spin, big);
		(void)hipEventRecord(e1, s);
		(void)hipStreamSynchronize(s);
		(void)hipEventElapsedTime(&ms, e0, e1);
	}
	return ms * 1e3f / N;
}
int main() {
	unsigned *a, *b;
	(void)hipMalloc(&a, 64 << 20); (void)hipMalloc(&b, 64 << 20);
	hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	(void)hipFuncSetAttribute((const void*)k<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for (int spin : {0, 1000}) {
		printf("spin %d: plain %.2f  load16KB %.2f  store16KB %.2f  load+store %.2f  load+store+bigargs %.2f us per launch\n", spin,
		       run<false, false, false>(s, a, b, spin, e0, e1), run<true, false, false>(s, a, b, spin, e0, e1),
		       run<false, true, false>(s, a, b, spin, e0, e1), run<true, true, false>(s, a, b, spin, e0, e1),
		       run<true, true, true>(s, a, b, spin, e0, e1));
	}
	return 0;
}
