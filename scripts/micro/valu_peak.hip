// Aggregate VALU issue rate of a SIMD on gfx950 against waves per SIMD, by wall clock (HIP events), not s_memtime:
// every wave runs the same chain of N instructions; grid = 256 CUs x k workgroups of 256 threads (one wave per SIMD each).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void chain(unsigned int* sink, unsigned int seed, int iters) {
	unsigned int a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, e = a * 11u;
	unsigned long long qm = __ballot(threadIdx.x & seed);
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			if (MODE == 0) {   // four independent chains of adds
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(e)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(e));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(e)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(e));
			}
			if (MODE == 1) {   // the cell arithmetic of a slot-run column: sub, min3, add per cell, four cells
				asm volatile("v_sub_u32 %0, %1, %2\n\tv_min3_u32 %0, %0, %2, %1\n\tv_add_u32 %2, %2, %0" : "=&v"(e), "+v"(b), "+v"(a));
				asm volatile("v_sub_u32 %0, %1, %2\n\tv_min3_u32 %0, %0, %2, %1\n\tv_add_u32 %2, %2, %0" : "=&v"(e), "+v"(b), "+v"(c));
				asm volatile("v_sub_u32 %0, %1, %2\n\tv_min3_u32 %0, %0, %2, %1\n\tv_add_u32 %2, %2, %0" : "=&v"(e), "+v"(b), "+v"(d));
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(a));
			}
			if (MODE == 2) {   // abs-diff accumulate: the Y-form column, one instruction per cell
				asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a) : "v"(e), "s"(seed));
				asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(b) : "v"(e), "s"(seed));
				asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(c) : "v"(e), "s"(seed));
				asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(d) : "v"(e), "s"(seed));
			}
			if (MODE == 3) {   // an ending read's cell: borrow of (mine - other - q) into an SGPR pair, shifted into the record, max
				unsigned long long bo, junk; unsigned int dummy;
				asm volatile("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(dummy), "=s"(bo) : "v"(a), "v"(b), "s"(qm));
				asm volatile("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "+v"(e), "=s"(junk) : "v"(e), "s"(bo));
				asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(b));
				asm volatile("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(dummy), "=s"(bo) : "v"(c), "v"(d), "s"(qm));
				asm volatile("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "+v"(e), "=s"(junk) : "v"(e), "s"(bo));
				asm volatile("v_max_u32 %0, %0, %1" : "+v"(c) : "v"(d));
			}
		}
	}
	sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e;
}
int main() {
	unsigned int* sink;
	hipMalloc(&sink, 256 * 16 * 256 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 4096;
	for (int mode = 0; mode < 4; ++mode)
		for (int k : {1, 2, 3, 4, 5, 6, 8}) {
			float best = 1e9f;
			for (int rep = 0; rep < 3; ++rep) {
				hipEventRecord(e0, 0);
				if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(256 * k), dim3(256), 0, 0, sink, 1u, iters);
				else if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(256 * k), dim3(256), 0, 0, sink, 1u, iters);
				else if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(256 * k), dim3(256), 0, 0, sink, 1u, iters);
				else hipLaunchKernelGGL(chain<3>, dim3(256 * k), dim3(256), 0, 0, sink, 1u, iters);
				hipEventRecord(e1, 0);
				hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1);
				if (ms < best) best = ms;
			}
			const double n_inst = (double)iters * 16 * (mode == 1 ? 10 : (mode == 3 ? 6 : 4));   // per wave
			const double per_simd = n_inst * k;                                 // k waves per SIMD
			printf("mode %d  %d waves/SIMD  %.3f ms  -> %.2f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz; one wave: %.2f cycles per instruction\n", mode, k, best,
			       best * 1e6 / per_simd, best * 1e6 / per_simd * 2.4, best * 1e6 / n_inst * 2.4);
		}
	return 0;
}
