// Round-6 probe: what a run boundary costs when a NARROW table (8 workgroups of 512 threads) stays in ONE launch -- a barrier among its workgroups with
// agent-scope release / acquire (correct whatever XCD the workgroups land on) -- against one launch per step (the r5 result: 2.26 us per dependent launch
// when the workgroups are packed onto one XCD).  Every step: each workgroup reads 8 KB another workgroup wrote in the previous step (16 bytes per thread),
// does a fixed block of arithmetic, writes its own 8 KB.  Variants: G independent groups of 8 workgroups, each with its own counter (tables at their own pace).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/r6pb scripts/micro/r6_persistent_barrier.hip && /tmp/r6pb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned work(unsigned a, unsigned b, unsigned iters) {
	for (unsigned i = 0; i < iters; ++i) {
#pragma unroll
		for (int j = 0; j < 16; ++j) a = a * b + 0x9E3779B9u;
	}
	return a;
}

// one launch per step (the baseline): `pack` as in slot_runx -- eight times the grid, every eighth workgroup works
__global__ __launch_bounds__(512) void step_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, unsigned wgs, unsigned iters, unsigned pack) {
	if (pack && (blockIdx.x & 7u)) return;
	const unsigned w = pack ? blockIdx.x >> 3 : blockIdx.x, group = w / wgs, me = w % wgs;
	const uint4 v = in[(group * wgs + (me * 5u + 3u) % wgs) * 512u + threadIdx.x];
	uint4 r = v;
	r.x = work(v.x, v.y | 5u, iters);
	out[w * 512u + threadIdx.x] = r;
}

// persistent: all steps in one launch.  MODE 0: plain stores + agent release fence + relaxed counter + agent acquire fence; MODE 1: write-through (sc1) payload
// stores, drained, relaxed counter, agent acquire fence.
template <int MODE>
__global__ __launch_bounds__(512) void walk_kernel(uint4* __restrict__ a, uint4* __restrict__ b, unsigned* __restrict__ counters, unsigned* __restrict__ fail, unsigned wgs,
                                                    unsigned iters, unsigned steps, unsigned pack) {
	if (pack && (blockIdx.x & 7u)) return;
	const unsigned w = pack ? blockIdx.x >> 3 : blockIdx.x, group = w / wgs, me = w % wgs;
	unsigned* counter = counters + group * 32u;   // (a 128-byte line per group)
	for (unsigned s = 0; s < steps; ++s) {
		const uint4* in = (s & 1u) ? b : a;
		uint4* out = (s & 1u) ? a : b;
		const uint4 v = in[(group * wgs + (me * 5u + 3u) % wgs) * 512u + threadIdx.x];
		uint4 r = v;
		r.x = work(v.x, v.y | 5u, iters);
		if (MODE == 0) out[w * 512u + threadIdx.x] = r;
		else {
			uint4* dst = out + w * 512u + threadIdx.x;
			typedef unsigned v4u __attribute__((ext_vector_type(4)));
			const v4u rv = {r.x, r.y, r.z, r.w};
			asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(rv) : "memory");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const unsigned want = (s + 1u) * wgs;
			unsigned spins = 0;
			while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
				__builtin_amdgcn_s_sleep(1);
				if (++spins > (1u << 22)) { *fail = 1u; break; }   // bounded: a lost workgroup must not hang the box
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		}
		__syncthreads();
	}
}

int main() {
	const unsigned steps = 2000;
	hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	unsigned *counters, *fail;
	(void)hipMalloc(&counters, 4096 * 4); (void)hipMalloc(&fail, 4);
	for (unsigned groups : {1u, 24u, 96u}) {
		for (unsigned wgs : {8u, 32u}) {
			if (groups * wgs > 768u) continue;   // (everything resident: three 512-thread workgroups per CU at most here)
			const size_t cells = (size_t)groups * wgs * 512;
			uint4 *a, *b;
			(void)hipMalloc(&a, cells * 16); (void)hipMalloc(&b, cells * 16);
			(void)hipMemset(a, 1, cells * 16); (void)hipMemset(b, 1, cells * 16);
			for (unsigned pack : {0u, 1u}) {
				if (pack && groups * wgs > 32u) continue;
				const dim3 grid(groups * wgs * (pack ? 8u : 1u));
				float per[2][3] = {};
				for (int rep = 0; rep < 2; ++rep) {
					const unsigned iters = rep ? 80u : 40u;
					float ms = 0;
					(void)hipEventRecord(e0, st);
					for (unsigned s = 0; s < steps; ++s) hipLaunchKernelGGL(step_kernel, grid, dim3(512), 0, st, (s & 1u) ? b : a, (s & 1u) ? a : b, wgs, iters, pack);
					(void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st); (void)hipEventElapsedTime(&ms, e0, e1);
					per[rep][0] = ms * 1e3f / steps;
					for (int mode = 0; mode < 2; ++mode) {
						(void)hipMemsetAsync(counters, 0, 4096 * 4, st); (void)hipMemsetAsync(fail, 0, 4, st);
						(void)hipEventRecord(e0, st);
						if (mode == 0) hipLaunchKernelGGL(walk_kernel<0>, grid, dim3(512), 0, st, a, b, counters, fail, wgs, iters, steps, pack);
						else hipLaunchKernelGGL(walk_kernel<1>, grid, dim3(512), 0, st, a, b, counters, fail, wgs, iters, steps, pack);
						(void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st); (void)hipEventElapsedTime(&ms, e0, e1);
						per[rep][1 + mode] = ms * 1e3f / steps;
						unsigned f = 0; (void)hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
						if (f) printf("  (a spin ran out: mode %d)\n", mode);
					}
				}
				printf("%3u group(s) x %2u workgroups%s: per step  launches %.2f / %.2f us   walk, release fence %.2f / %.2f us   walk, write-through stores %.2f / %.2f us   (40 / 80 trips of arithmetic)\n",
				       groups, wgs, pack ? ", packed onto one XCD" : "", per[0][0], per[1][0], per[0][1], per[1][1], per[0][2], per[1][2]);
			}
			(void)hipFree(a); (void)hipFree(b);
		}
	}
	return 0;
}
