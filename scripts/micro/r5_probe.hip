// Round-5 probe (one MI355X): what a hand-scheduled column loop can count on.  Wall-clock based (hipEvents over long kernels) so that no
// assumption about the tick of s_memtime enters; the s_memtime delta of the same region is printed next to it as its calibration.
//   A  issue interval of ONE wave: dependent v_add chain, four independent v_sad_u32 chains (VGPR / SGPR second operand), the "plain column"
//      pattern 4 x v_sad + s_and + s_cbranch, at 1, 2 and 4 waves per SIMD
//   B  instruction cache across launches: the same instruction stream as straight-line code (24 KB, 96 KB) and as a loop, launched back to back
//   C  latencies inside one wave (dependent repetitions): 4 x ds_bpermute, ds_read_b128 from a uniform address, s_load_dwordx16 (scalar cache hit),
//      v_readlane -> VALU use, v_cmp -> SGPR pair -> v_subb, workgroup barrier with 8 waves, LDS exchange (write b128, barrier, read b128)
// Build: hipcc --offload-arch=gfx950 -O3 -o r5_probe r5_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))

static __device__ __forceinline__ unsigned long long memtime() {
	unsigned long long t;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
	return t;
}

// ---- A
template <int MODE>
__global__ void issue_probe(unsigned long long* ticks, unsigned* sink, unsigned iters, unsigned k0) {
	unsigned d0 = threadIdx.x, d1 = d0 * 3u, d2 = d0 * 5u, d3 = d0 * 7u, x = d0 * 11u + 0x80000000u, kv = k0 + 0x80000000u;
	unsigned ks = k0 + 0x80000001u, cw = k0 & 0u;   // cw == 0: the branch below is never taken
	const unsigned long long t0 = memtime();
	for (unsigned it = 0; it < iters; ++it) {
		if (MODE == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(d0) : "v"(x));) }
		if (MODE == 1) { REP16(asm volatile("v_sad_u32 %0, %4, %5, %0\n\tv_sad_u32 %1, %4, %5, %1\n\tv_sad_u32 %2, %4, %5, %2\n\tv_sad_u32 %3, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x), "v"(kv));) }
		if (MODE == 2) { REP16(asm volatile("v_sad_u32 %0, %4, %5, %0\n\tv_sad_u32 %1, %4, %5, %1\n\tv_sad_u32 %2, %4, %5, %2\n\tv_sad_u32 %3, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x), "s"(ks));) }
		if (MODE == 3) {   // the plain column: four cells + the control test (6 instructions), 16 columns per trip
			REP16(asm volatile("v_sad_u32 %0, %4, %5, %0\n\tv_sad_u32 %1, %4, %5, %1\n\tv_sad_u32 %2, %4, %5, %2\n\tv_sad_u32 %3, %4, %5, %3\n\t"
			                   "s_and_b32 s4, %6, 0x8003\n\ts_cbranch_scc1 1f\n1:" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x), "v"(kv), "s"(cw) : "s4", "scc");)
		}
		if (MODE == 4) {   // ... with the line of the column after next requested from LDS and waited for with a count (8 instructions)
			REP16(asm volatile("ds_read_b128 v[40:43], %6\n\ts_waitcnt lgkmcnt(1)\n\t"
			                   "v_sad_u32 %0, %4, %5, %0\n\tv_sad_u32 %1, %4, %5, %1\n\tv_sad_u32 %2, %4, %5, %2\n\tv_sad_u32 %3, %4, %5, %3\n\t"
			                   "s_and_b32 s4, %7, 0x8003\n\ts_cbranch_scc1 1f\n1:" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x), "v"(kv), "v"(k0 & 0u), "s"(cw) : "s4", "scc", "v40", "v41", "v42", "v43", "memory");)
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
	}
	const unsigned long long t1 = memtime();
	if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
	sink[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3;
}

// ---- B: N "columns" of straight-line code, or the same count as a loop over a 16-column body
template <int COLS256>   // straight-line columns / 256
__global__ void straight(unsigned* sink, unsigned k0) {
	unsigned d0 = threadIdx.x, d1 = d0 * 3u, d2 = d0 * 5u, d3 = d0 * 7u, x = d0 * 11u + 0x80000000u, kv = k0 + 0x80000000u, cw = k0 & 0u;
#define COL asm volatile("v_sad_u32 %0, %4, %5, %0\n\tv_sad_u32 %1, %4, %5, %1\n\tv_sad_u32 %2, %4, %5, %2\n\tv_sad_u32 %3, %4, %5, %3\n\ts_and_b32 s4, %6, 0x8003\n\ts_cbranch_scc1 1f\n1:" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x), "v"(kv), "s"(cw) : "s4", "scc");
	if (COLS256 >= 1) { REP256(COL) }
	if (COLS256 >= 2) { REP256(COL) }
	if (COLS256 >= 4) { REP256(COL) REP256(COL) }
	if (COLS256 >= 8) { REP256(COL) REP256(COL) REP256(COL) REP256(COL) }
	sink[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3;
}
__global__ void looped(unsigned* sink, unsigned k0, unsigned trips) {
	unsigned d0 = threadIdx.x, d1 = d0 * 3u, d2 = d0 * 5u, d3 = d0 * 7u, x = d0 * 11u + 0x80000000u, kv = k0 + 0x80000000u, cw = k0 & 0u;
	for (unsigned t = 0; t < trips; ++t) { REP16(COL) }
	sink[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3;
}

// ---- C
template <int MODE>
__global__ __launch_bounds__(512) void latency_probe(unsigned long long* ticks, unsigned* sink, const unsigned* table, unsigned zero) {
	__shared__ __attribute__((aligned(16))) unsigned lds[4096];
	const unsigned tid = threadIdx.x, lane = tid & 63u;
	for (unsigned i = tid; i < 4096; i += blockDim.x) lds[i] = i + zero;
	__syncthreads();
	unsigned d0 = tid, d1 = tid * 3u, d2 = tid * 5u, d3 = tid * 7u, acc = 0;
	const unsigned addr = ((lane ^ 4u) << 2) + zero;
	const unsigned long long t0 = memtime();
	if (MODE == 0) {   // 4 x ds_bpermute, waited for, results fed back (what a lane-slot ending does)
		REP64(asm volatile("ds_bpermute_b32 v40, %4, %0\n\tds_bpermute_b32 v41, %4, %1\n\tds_bpermute_b32 v42, %4, %2\n\tds_bpermute_b32 v43, %4, %3\n\ts_waitcnt lgkmcnt(0)\n\t"
		                   "v_max_u32 %0, %0, v40\n\tv_max_u32 %1, %1, v41\n\tv_max_u32 %2, %2, v42\n\tv_max_u32 %3, %3, v43" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(addr) : "v40", "v41", "v42", "v43", "memory");)
	}
	if (MODE == 1) {   // ds_read_b128 from a wave-uniform address, address depends on the previous result
		unsigned a = zero;
		REP64(asm volatile("ds_read_b128 v[40:43], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0x30, v40" : "+v"(a) :: "v40", "v41", "v42", "v43", "memory");)
		acc = a;
	}
	if (MODE == 2) {   // s_load_dwordx16 from the scalar cache (same 64 bytes again and again), offset depends on the previous result
		unsigned off = zero;
		REP64(asm volatile("s_load_dwordx16 s[36:51], %1, %0\n\ts_waitcnt lgkmcnt(0)\n\ts_and_b32 %0, s36, 0x40" : "+s"(off) : "s"(table) : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "memory");)
		acc = off;
	}
	if (MODE == 3) {   // v_readlane -> SGPR -> VALU operand
		REP64(asm volatile("v_readlane_b32 s36, %0, 3\n\tv_add_u32 %0, s36, %0" : "+v"(d0) :: "s36");)
	}
	if (MODE == 4) {   // v_cmp -> SGPR pair -> v_subb carry-in -> v_addc (the tie decision of one cell)
		REP64(asm volatile("v_cmp_ne_u32 s[36:37], 0, %0\n\tv_subb_co_u32 v40, s[38:39], %0, %1, s[36:37]\n\tv_addc_co_u32 %0, s[38:39], %0, %0, s[38:39]" : "+v"(d0) : "v"(d1) : "s36", "s37", "s38", "s39", "v40");)
	}
	if (MODE == 5) {   // workgroup barrier alone
		REP64(asm volatile("s_barrier" ::: "memory");)
	}
	if (MODE == 6) {   // the wave-slot exchange: 16 bytes per thread through LDS, barrier, the partner wave's 16 bytes
		const unsigned mine = tid * 16u + zero, theirs = (tid ^ 64u) * 16u + zero;
		REP64(asm volatile("ds_write_b128 %4, v[44:47]\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\tds_read_b128 v[40:43], %5\n\ts_waitcnt lgkmcnt(0)\n\t"
		                   "v_max_u32 %0, %0, v40\n\tv_max_u32 %1, %1, v41\n\tv_max_u32 %2, %2, v42\n\tv_max_u32 %3, %3, v43\n\ts_barrier"
		                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(mine), "v"(theirs) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");)
	}
	const unsigned long long t1 = memtime();
	if (tid == 0) ticks[blockIdx.x] = t1 - t0;
	sink[blockIdx.x * blockDim.x + tid] = d0 + d1 + d2 + d3 + acc;
}

static double mean(const std::vector<unsigned long long>& v) { double s = 0; for (auto x : v) s += (double)x; return s / v.size(); }

int main() {
	unsigned long long* ticks; unsigned *sink, *table;
	(void)hipMalloc(&ticks, 1024 * 8); (void)hipMalloc(&sink, 1024 * 1024 * 4); (void)hipMalloc(&table, 4096);
	(void)hipMemset(table, 0, 4096);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	std::vector<unsigned long long> h(256);
	// ---- A
	const char* names[5] = {"dependent v_add chain (64 per trip)", "4 independent v_sad, VGPR operand", "4 independent v_sad, SGPR operand", "plain column: 4 v_sad + s_and + s_cbranch",
	                        "plain column + ds_read_b128 + counted wait"};
	const int per_trip[5] = {64, 64, 64, 96, 128};
	for (int threads : {256, 512, 1024}) {
		for (int mode = 0; mode < 5; ++mode) {
			const unsigned iters = 20000;
			float ms = 0;
			for (int rep = 0; rep < 2; ++rep) {
				(void)hipEventRecord(e0, 0);
				if (mode == 0) hipLaunchKernelGGL(issue_probe<0>, dim3(256), dim3(threads), 0, 0, ticks, sink, iters, 1u);
				if (mode == 1) hipLaunchKernelGGL(issue_probe<1>, dim3(256), dim3(threads), 0, 0, ticks, sink, iters, 1u);
				if (mode == 2) hipLaunchKernelGGL(issue_probe<2>, dim3(256), dim3(threads), 0, 0, ticks, sink, iters, 1u);
				if (mode == 3) hipLaunchKernelGGL(issue_probe<3>, dim3(256), dim3(threads), 0, 0, ticks, sink, iters, 1u);
				if (mode == 4) hipLaunchKernelGGL(issue_probe<4>, dim3(256), dim3(threads), 0, 0, ticks, sink, iters, 1u);
				(void)hipEventRecord(e1, 0);
				(void)hipEventSynchronize(e1);
				(void)hipEventElapsedTime(&ms, e0, e1);
			}
			(void)hipMemcpy(h.data(), ticks, 256 * 8, hipMemcpyDeviceToHost);
			const double n = (double)iters * per_trip[mode];
			printf("A %4d threads/WG (%d wave(s)/SIMD)  %-44s %7.3f ns per instruction of one wave (wall), %7.3f memtime ticks per instruction, %.4f ticks/ns\n", threads, threads / 256,
			       names[mode], ms * 1e6 / n, mean(h) / n, mean(h) / (ms * 1e6));
		}
	}
	// ---- B
	for (int threads : {512}) {
		float ms[6] = {0};
		const int launches = 300;
		for (int variant = 0; variant < 6; ++variant) {
			for (int rep = 0; rep < 2; ++rep) {
				(void)hipEventRecord(e0, 0);
				for (int i = 0; i < launches; ++i) {
					if (variant == 0) hipLaunchKernelGGL(straight<2>, dim3(256), dim3(threads), 0, 0, sink, 1u);        // 512 columns ~ 26 KB of code
					if (variant == 1) hipLaunchKernelGGL(looped, dim3(256), dim3(threads), 0, 0, sink, 1u, 32u);
					if (variant == 2) hipLaunchKernelGGL(straight<8>, dim3(256), dim3(threads), 0, 0, sink, 1u);        // 2048 columns ~ 104 KB of code
					if (variant == 3) hipLaunchKernelGGL(looped, dim3(256), dim3(threads), 0, 0, sink, 1u, 128u);
					if (variant == 4) hipLaunchKernelGGL(straight<1>, dim3(256), dim3(threads), 0, 0, sink, 1u);        // 256 columns ~ 13 KB
					if (variant == 5) hipLaunchKernelGGL(looped, dim3(256), dim3(threads), 0, 0, sink, 1u, 16u);
				}
				(void)hipEventRecord(e1, 0);
				(void)hipEventSynchronize(e1);
				(void)hipEventElapsedTime(&ms[variant], e0, e1);
			}
		}
		printf("B %d threads/WG, 256 WGs, %d back-to-back launches: 512 columns straight-line %.2f us / looped %.2f us per launch; 2048 columns straight %.2f / looped %.2f; 256 columns straight %.2f / looped %.2f\n",
		       threads, launches, ms[0] * 1e3 / launches, ms[1] * 1e3 / launches, ms[2] * 1e3 / launches, ms[3] * 1e3 / launches, ms[4] * 1e3 / launches, ms[5] * 1e3 / launches);
	}
	// ---- C
	const char* cn[7] = {"4 x ds_bpermute + wait + 4 v_max", "ds_read_b128 (uniform address) + wait + v_and", "s_load_dwordx16 (scalar cache) + wait + s_and", "v_readlane -> v_add",
	                     "v_cmp -> v_subb -> v_addc", "s_barrier (8 waves)", "LDS exchange: write b128, barrier, read b128, 4 v_max, barrier"};
	for (int mode = 0; mode < 7; ++mode) {
		for (int rep = 0; rep < 2; ++rep) {
			if (mode == 0) hipLaunchKernelGGL(latency_probe<0>, dim3(256), dim3(512), 0, 0, ticks, sink, table, 0u);
			if (mode == 1) hipLaunchKernelGGL(latency_probe<1>, dim3(256), dim3(512), 0, 0, ticks, sink, table, 0u);
			if (mode == 2) hipLaunchKernelGGL(latency_probe<2>, dim3(256), dim3(512), 0, 0, ticks, sink, table, 0u);
			if (mode == 3) hipLaunchKernelGGL(latency_probe<3>, dim3(256), dim3(512), 0, 0, ticks, sink, table, 0u);
			if (mode == 4) hipLaunchKernelGGL(latency_probe<4>, dim3(256), dim3(512), 0, 0, ticks, sink, table, 0u);
			if (mode == 5) hipLaunchKernelGGL(latency_probe<5>, dim3(256), dim3(512), 0, 0, ticks, sink, table, 0u);
			if (mode == 6) hipLaunchKernelGGL(latency_probe<6>, dim3(256), dim3(512), 0, 0, ticks, sink, table, 0u);
			(void)hipDeviceSynchronize();
		}
		(void)hipMemcpy(h.data(), ticks, 256 * 8, hipMemcpyDeviceToHost);
		printf("C 512 threads/WG  %-62s %8.1f memtime ticks per repetition (wave 0 of each WG, 64 dependent repetitions)\n", cn[mode], mean(h) / 64.0);
	}
	return 0;
}
