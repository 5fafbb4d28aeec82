// Round-5 probe 2: what the boundary between two DEPENDENT launches of a run kernel consists of.  256 workgroups x 512 threads, a fixed block of
// arithmetic (~4.6 us), back-to-back launches on one stream, with / without: 90 KB of dynamic LDS, a 512-byte kernel argument, 16 bytes per thread read
// from what the previous launch wrote (scattered over the other workgroups: crosses the XCDs), 16 bytes per thread + 11 record bytes stored at the end.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { unsigned w[120]; };
template <bool READ, bool WRITE, bool REC, bool DEP = false, bool SAMEXCD = false>
__global__ __launch_bounds__(512) void k(Big big, const uint4* __restrict__ in, uint4* __restrict__ out, unsigned char* __restrict__ rec, unsigned iters) {
	extern __shared__ unsigned sm[];
	const unsigned gid = blockIdx.x * 512u + threadIdx.x;
	uint4 v = make_uint4(gid, big.w[7], big.w[100], 3u);
	if (READ) v = in[(SAMEXCD ? ((blockIdx.x + 40u) & 255u) : ((blockIdx.x * 37u + 11u) & 255u)) * 512u + threadIdx.x];   // (SAMEXCD: a block written by a workgroup of the SAME XCD: workgroup b runs on XCD b mod 8)
   // another workgroup's 8 KB block (contiguous, like a run's entering cells): written by the previous launch
	unsigned a = DEP ? v.x : gid, b = big.w[3] | 5u;   // (DEP: the arithmetic starts from what was read -- the latency is exposed)
	for (unsigned i = 0; i < iters; ++i) {
#pragma unroll
		for (int j = 0; j < 16; ++j) a = a * b + 0x9E3779B9u;
	}
	v.x ^= a;
	if (WRITE) out[gid] = v;
	if (REC) for (int e = 0; e < 11; ++e) rec[(size_t)(blockIdx.x * 11 + e) * 512 + threadIdx.x] = (unsigned char)(a >> e);
	if (a == 0x12345u) sm[threadIdx.x] = a;
}
int main() {
	uint4 *a, *b; unsigned char* rec;
	for (int alloc = 0; alloc < 3; ++alloc) {
	if (alloc == 0) { (void)hipMalloc(&a, 131072 * 16); (void)hipMalloc(&b, 131072 * 16); printf("== exchange buffers: hipMalloc (coarse-grained)\n"); }
	if (alloc == 1) { if (hipExtMallocWithFlags((void**)&a, 131072 * 16, hipDeviceMallocUncached) != hipSuccess || hipExtMallocWithFlags((void**)&b, 131072 * 16, hipDeviceMallocUncached) != hipSuccess) { printf("uncached allocation failed\n"); continue; } printf("== exchange buffers: hipDeviceMallocUncached\n"); }
	if (alloc == 2) { if (hipExtMallocWithFlags((void**)&a, 131072 * 16, hipDeviceMallocFinegrained) != hipSuccess || hipExtMallocWithFlags((void**)&b, 131072 * 16, hipDeviceMallocFinegrained) != hipSuccess) { printf("fine-grained allocation failed\n"); continue; } printf("== exchange buffers: hipDeviceMallocFinegrained\n"); }
	(void)hipMalloc(&rec, 256 * 11 * 512);
	(void)hipMemset(a, 1, 131072 * 16); (void)hipMemset(b, 1, 131072 * 16);
	hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	(void)hipFuncSetAttribute((const void*)k<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k<true, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	Big big{};
	const int N = 2000;
	const char* names[9] = {"compute only, no LDS", "compute only, 90 KB LDS", "+ read", "+ write", "+ read + write", "+ read + write + records", "+ read (needed at once) + write + records", "+ read (needed at once), nothing written", "+ read (needed at once, written on the SAME XCD) + write + records"};
	for (int variant = (alloc ? 4 : 0); variant < 9; ++variant) {
		const size_t lds = variant == 0 ? 0 : 90 * 1024;
		float ms = 0, per[2] = {0, 0};
		for (int rep = 0; rep < 4; ++rep) {
			const unsigned iters = (rep & 1) ? 80u : 40u;
			(void)hipEventRecord(e0, s);
			for (int i = 0; i < N; ++i) {
				const uint4* in = (i & 1) ? a : b; uint4* out = (i & 1) ? b : a;
				if (variant <= 1) hipLaunchKernelGGL((k<false, false, false>), dim3(256), dim3(512), lds, s, big, in, out, rec, iters);
				if (variant == 2) hipLaunchKernelGGL((k<true, false, false>), dim3(256), dim3(512), lds, s, big, in, out, rec, iters);
				if (variant == 3) hipLaunchKernelGGL((k<false, true, false>), dim3(256), dim3(512), lds, s, big, in, out, rec, iters);
				if (variant == 4) hipLaunchKernelGGL((k<true, true, false>), dim3(256), dim3(512), lds, s, big, in, out, rec, iters);
				if (variant == 5) hipLaunchKernelGGL((k<true, true, true>), dim3(256), dim3(512), lds, s, big, in, out, rec, iters);
				if (variant == 6) hipLaunchKernelGGL((k<true, true, true, true>), dim3(256), dim3(512), lds, s, big, in, out, rec, iters);
				if (variant == 7) hipLaunchKernelGGL((k<true, false, false, true>), dim3(256), dim3(512), lds, s, big, in, out, rec, iters);
				if (variant == 8) hipLaunchKernelGGL((k<true, true, true, true, true>), dim3(256), dim3(512), lds, s, big, in, out, rec, iters);
			}
			(void)hipEventRecord(e1, s);
			(void)hipStreamSynchronize(s);
			(void)hipEventElapsedTime(&ms, e0, e1);
			per[rep & 1] = ms * 1e3f / N;
		}
		printf("%-50s %.3f us per launch with 40 trips, %.3f with 80: fixed part %.3f us, arithmetic %.3f us per 40 trips\n", names[variant], per[0], per[1], 2 * per[0] - per[1], per[1] - per[0]);
	}
	}
	return 0;
}
