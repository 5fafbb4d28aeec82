// Microbenchmark 3: how long after its first instruction can a wave USE a kernel argument, and when does data loaded
// through a pointer argument arrive?  Built twice: plain, and with -mllvm -amdgpu-kernarg-preload-count=14 (gfx950 can
// deliver the leading scalar/pointer arguments in SGPRs at wave launch).  Chain of dependent launches, 256 x 1024.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { unsigned a[90]; };
__global__ __launch_bounds__(1024) void k(const unsigned* p, unsigned* q, unsigned long long* stamp, unsigned n, Big big) {
	const unsigned long long t0 = __builtin_readcyclecounter();
	unsigned v = n;
	asm volatile("s_nop 0" :: "s"(v));  // first use of a (preloadable) argument
	const unsigned long long t1 = __builtin_readcyclecounter();
	unsigned x = p[(blockIdx.x * 1024 + threadIdx.x) & (n - 1)];
	asm volatile("v_nop" :: "v"(x));
	const unsigned long long t2 = __builtin_readcyclecounter();
	unsigned y = big.a[threadIdx.x % 90];  // argument beyond the preloaded ones
	asm volatile("v_nop" :: "v"(y));
	const unsigned long long t3 = __builtin_readcyclecounter();
	q[blockIdx.x * 1024 + threadIdx.x] = x + y;
	if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd(&stamp[0], t1 - t0); atomicAdd(&stamp[1], t2 - t0); atomicAdd(&stamp[2], t3 - t0); }
}
int main() {
	unsigned *a, *b; unsigned long long* st;
	(void)hipMalloc(&a, 4 << 20); (void)hipMalloc(&b, 4 << 20); (void)hipMalloc(&st, 64);
	(void)hipMemset(st, 0, 64);
	hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	Big big{};
	const int N = 2000;
	for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, st, 1u << 18, big);
	(void)hipStreamSynchronize(s);
	unsigned long long h[3];
	(void)hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
	printf("cycles after the first instruction (wave 0 of workgroup 0, mean of %d launches): argument usable %.0f, loaded data %.0f, late argument %.0f\n",
	       N, (double)h[0] / N, (double)h[1] / N, (double)h[2] / N);
	return 0;
}
