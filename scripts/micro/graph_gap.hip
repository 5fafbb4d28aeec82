// Microbenchmark 4: does a captured hipGraph shorten the boundary between dependent kernels compared with plain
// stream launches?  256 x 1024 threads, 112 KiB dynamic LDS (one workgroup per CU, like the run kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned* p, const unsigned* q, int spin) {
	extern __shared__ unsigned sm[];
	unsigned v = q[blockIdx.x * 1024 + threadIdx.x];
	for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;
	p[blockIdx.x * 1024 + threadIdx.x] = v;
	if (v == 0x12345678u) p[0] = sm[threadIdx.x];
}
int main() {
	unsigned *a, *b;
	(void)hipMalloc(&a, 4 << 20); (void)hipMalloc(&b, 4 << 20);
	hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	(void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int N = 2000;
	for (int spin : {0, 2000}) {
		float ms_stream = 0, ms_graph = 0;
		for (int rep = 0; rep < 2; ++rep) {
			(void)hipEventRecord(e0, s);
			for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 112 * 1024, s, (i & 1) ? a : b, (i & 1) ? b : a, spin);
			(void)hipEventRecord(e1, s);
			(void)hipStreamSynchronize(s);
			(void)hipEventElapsedTime(&ms_stream, e0, e1);
		}
		hipGraph_t g; hipGraphExec_t ge;
		(void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
		for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 112 * 1024, s, (i & 1) ? a : b, (i & 1) ? b : a, spin);
		(void)hipStreamEndCapture(s, &g);
		if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
		for (int rep = 0; rep < 3; ++rep) {
			(void)hipEventRecord(e0, s);
			(void)hipGraphLaunch(ge, s);
			(void)hipEventRecord(e1, s);
			(void)hipStreamSynchronize(s);
			(void)hipEventElapsedTime(&ms_graph, e0, e1);
		}
		printf("spin %d: stream %.2f us per launch, graph %.2f us per launch\n", spin, ms_stream * 1e3f / N, ms_graph * 1e3f / N);
		(void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
	}
	return 0;
}
