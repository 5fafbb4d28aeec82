// Issue-rate probe for gfx950: cycles per instruction of ONE wave for dependent / independent VALU chains, SALU, and
// VALU+SALU mixes, with 1, 2 and 4 waves per SIMD (how much of a latency-bound column chain is instruction issue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void probe(unsigned long long* out, unsigned int* sink, unsigned int seed) {
	unsigned int a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u;
	unsigned int s = seed, s2 = seed * 3u;
	unsigned long long t0, t1;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
	for (int it = 0; it < 64; ++it) {
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			if (MODE == 0) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(c));
			                 asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(d)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
			if (MODE == 1) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(d)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(d));
			                 asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d) : "v"(a)); }
			if (MODE == 2) { asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(s2)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(s2));
			                 asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(s2)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(s2)); }
			if (MODE == 3) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(s2));
			                 asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(s)); }
			if (MODE == 4) { asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a) : "v"(d));
			                 asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a) : "v"(d)); }
		}
	}
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
	if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
	sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + s + s2;
}
int main() {
	unsigned long long* out; unsigned int* sink;
	hipMalloc(&out, 4096 * 8); hipMalloc(&sink, 4096 * 1024 * 4);
	const char* names[5] = {"dependent v_add chain", "3 independent v_add + xor", "dependent s_add chain", "VALU/SALU alternating", "dependent min3/sub chain"};
	for (int grid : {256, 512}) for (int threads : {64, 256, 512, 1024}) {
		for (int mode = 0; mode < 5; ++mode) {
			for (int rep = 0; rep < 2; ++rep) {
				if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(threads), 0, 0, out, sink, 1u);
				if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(threads), 0, 0, out, sink, 1u);
				if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(threads), 0, 0, out, sink, 1u);
				if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(threads), 0, 0, out, sink, 1u);
				if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(threads), 0, 0, out, sink, 1u);
				hipDeviceSynchronize();
			}
			std::vector<unsigned long long> h(256);
			hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
			double sum = 0; for (auto v : h) sum += (double)v;
			printf("grid %d %4d threads/WG (%d waves/SIMD per WG)  %-28s %.2f cycles per instruction (wave 0 of each WG)\n", grid, threads, threads / 256 ? threads / 256 : 1, names[mode], sum / 256 / (64 * 16 * 4));
		}
	}
	return 0;
}
