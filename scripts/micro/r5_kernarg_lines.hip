// Round-5 probe 3: which part of the kernel-argument segment is at hand when a wave starts?  One kernel per 64-byte line L of a 512-byte argument:
// cycles from the wave's first instruction until a word of line L (and nothing before it) is usable.  Chain of dependent launches, 256 x 512.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { unsigned a[128]; };
template <int LINE>
__global__ __launch_bounds__(512) void k(Big big, unsigned* q, unsigned long long* stamp) {
	const unsigned long long t0 = __builtin_readcyclecounter();
	unsigned v = big.a[LINE * 16];
	asm volatile("s_nop 0" ::"s"(v));
	const unsigned long long t1 = __builtin_readcyclecounter();
	q[blockIdx.x * 512 + threadIdx.x] = v;
	if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stamp[LINE], t1 - t0);
}
int main() {
	unsigned* b; unsigned long long* st;
	(void)hipMalloc(&b, 4 << 20); (void)hipMalloc(&st, 64);
	(void)hipMemset(st, 0, 64);
	hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	Big big{};
	const int N = 1000;
	for (int i = 0; i < N; ++i) {
		hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, s, big, b, st);
		hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, s, big, b, st);
		hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, s, big, b, st);
		hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, s, big, b, st);
		hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, s, big, b, st);
		hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, s, big, b, st);
		hipLaunchKernelGGL(k<7>, dim3(256), dim3(512), 0, s, big, b, st);
	}
	(void)hipStreamSynchronize(s);
	unsigned long long h[8];
	(void)hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
	for (int l : {0, 1, 2, 3, 4, 6, 7}) printf("line %d of the kernel arguments (bytes %d ..): usable %.0f cycles after the wave's first instruction\n", l, l * 64, (double)h[l] / N);
	return 0;
}
