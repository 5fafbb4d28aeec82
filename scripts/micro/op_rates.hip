// Throughput of single VALU opcodes on gfx950 at 8 waves per SIMD (wall clock), four independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP4(TEXT, ...)                                                   \
	asm volatile(TEXT : "+v"(a) : "v"(e), "v"(f), "s"(seed), ##__VA_ARGS__); \
	asm volatile(TEXT : "+v"(b) : "v"(e), "v"(f), "s"(seed), ##__VA_ARGS__); \
	asm volatile(TEXT : "+v"(c) : "v"(e), "v"(f), "s"(seed), ##__VA_ARGS__); \
	asm volatile(TEXT : "+v"(d) : "v"(e), "v"(f), "s"(seed), ##__VA_ARGS__);
template <int MODE>
__global__ __launch_bounds__(256) void chain(unsigned int* sink, unsigned int seed, int iters) {
	unsigned int a = threadIdx.x + seed, b = a * 3u, c = a * 5u, d = a * 7u, e = a * 11u, f = a * 13u;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			if (MODE == 0) { OP4("v_add_u32 %0, %0, %1") }
			if (MODE == 1) { OP4("v_min3_u32 %0, %0, %1, %2") }
			if (MODE == 2) { OP4("v_sad_u32 %0, %1, %2, %0") }
			if (MODE == 3) { OP4("v_sad_u32 %0, %1, %3, %0") }
			if (MODE == 4) { OP4("v_sad_u32 %0, %0, %1, 0") }
			if (MODE == 5) { OP4("v_add3_u32 %0, %0, %1, %2") }
			if (MODE == 6) { OP4("v_max_u32 %0, %0, %1") }
			if (MODE == 7) { OP4("v_sub_u32 %0, %1, %0") }
			if (MODE == 8) { OP4("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
			if (MODE == 9) { OP4("v_add_u32 %0, %0, %3") }
			if (MODE == 10) { OP4("v_min_u32 %0, %0, %1") }
			if (MODE == 11) { OP4("v_xor_b32 %0, %0, %1") }
			if (MODE == 12) { OP4("v_lshl_add_u32 %0, %0, 1, %1") }
			if (MODE == 13) { OP4("v_cndmask_b32 %0, %0, %1, vcc") }
			if (MODE == 14) { OP4("v_bfe_u32 %0, %0, %1, 1") }
			if (MODE == 15) { OP4("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
			if (MODE == 16) { OP4("v_med3_i32 %0, %0, %1, %2") }
			if (MODE == 17) { OP4("v_max3_u32 %0, %0, %1, %2") }
		}
	}
	sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e;
}
template <int MODE>
float run(unsigned int* sink, int k, int iters) {
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 3; ++rep) {
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(chain<MODE>, dim3(256 * k), dim3(256), 0, 0, sink, 1u, iters);
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	return best;
}
int main() {
	unsigned int* sink;
	hipMalloc(&sink, 256 * 16 * 256 * 4);
	const int iters = 2048;
	const char* names[18] = {"v_add_u32", "v_min3_u32", "v_sad_u32 (3 vgpr)", "v_sad_u32 (sgpr src1)", "v_sad_u32 (src2 = 0)", "v_add3_u32", "v_max_u32", "v_sub_u32", "v_mov_b32_dpp quad_perm",
	                         "v_add_u32 (sgpr)", "v_min_u32", "v_xor_b32", "v_lshl_add_u32", "v_cndmask_b32 vcc", "v_bfe_u32", "v_add_u32_dpp", "v_med3_i32", "v_max3_u32"};
	for (int k : {2, 8}) {
		float ms[18];
		ms[0] = run<0>(sink, k, iters); ms[1] = run<1>(sink, k, iters); ms[2] = run<2>(sink, k, iters); ms[3] = run<3>(sink, k, iters); ms[4] = run<4>(sink, k, iters);
		ms[5] = run<5>(sink, k, iters); ms[6] = run<6>(sink, k, iters); ms[7] = run<7>(sink, k, iters); ms[8] = run<8>(sink, k, iters); ms[9] = run<9>(sink, k, iters);
		ms[10] = run<10>(sink, k, iters); ms[11] = run<11>(sink, k, iters); ms[12] = run<12>(sink, k, iters); ms[13] = run<13>(sink, k, iters); ms[14] = run<14>(sink, k, iters);
		ms[15] = run<15>(sink, k, iters); ms[16] = run<16>(sink, k, iters); ms[17] = run<17>(sink, k, iters);
		for (int m = 0; m < 18; ++m) printf("%d waves/SIMD  %-26s %.3f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", k, names[m], ms[m] * 1e6 / ((double)iters * 64 * k), ms[m] * 1e6 / ((double)iters * 64 * k) * 2.4);
	}
	return 0;
}
