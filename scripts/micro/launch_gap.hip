// Microbenchmark: cost of a chain of dependent launches as a function of launch shape (threads, dynamic LDS, work).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void k(unsigned* p, int spin) {
	extern __shared__ unsigned sm[];
	unsigned v = threadIdx.x;
	for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;
	if (spin < 0) sm[threadIdx.x] = v;
	if (v == 0x12345678u) p[blockIdx.x] = v + sm[0];
}
int main() {
	unsigned* d; hipMalloc(&d, 1 << 20);
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct Cfg { int grid, block, lds, spin; };
	std::vector<Cfg> cfgs = {{256, 1024, 112 * 1024, 0}, {256, 1024, 0, 0}, {256, 256, 0, 0}, {256, 512, 60 * 1024, 0}, {256, 1024, 112 * 1024, 2000},
	                         {256, 1024, 64 * 1024, 0}, {256, 1024, 80 * 1024, 0}, {256, 512, 112 * 1024, 0}, {8, 1024, 112 * 1024, 0}, {512, 512, 60 * 1024, 0}};
	for (auto c : cfgs) {
		for (int rep = 0; rep < 2; ++rep) {
			const int N = 2000;
			hipEventRecord(e0, s);
			for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(c.grid), dim3(c.block), c.lds, s, d, c.spin);
			hipEventRecord(e1, s);
			hipStreamSynchronize(s);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			if (rep) printf("grid %4d block %4d lds %6d spin %5d : %.2f us per launch\n", c.grid, c.block, c.lds, c.spin, ms * 1e3 / N);
		}
	}
	return 0;
}
