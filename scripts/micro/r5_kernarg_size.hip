// Round-5 probe: what a dependent launch costs as a function of the size of its kernel arguments and of its grid -- a group launch (slot_groupx) passes 2 KB of
// pointers and starts 768 workgroups; the same launches with a 64-byte argument / with the pointers fetched from a device array instead.
// Build: hipcc --offload-arch=gfx950 -O3 -o r5_kernarg_size r5_kernarg_size.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big { unsigned n, pad; const unsigned* entry[248]; };      // SlotGroupArgs
struct Small { unsigned n, pad; const unsigned* const* table; };  // the same pointers behind one pointer

template <int SPIN>
__global__ __launch_bounds__(512) void k_big(Big a, unsigned* out) {
	const unsigned t = blockIdx.x % a.n;
	unsigned v = a.entry[t][threadIdx.x & 63];
	for (int i = 0; i < SPIN; ++i) v = v * 1664525u + 1013904223u;
	if (threadIdx.x == 0) out[blockIdx.x] = v;
}
template <int SPIN>
__global__ __launch_bounds__(512) void k_small(Small a, unsigned* out) {
	const unsigned t = blockIdx.x % a.n;
	unsigned v = a.table[t][threadIdx.x & 63];
	for (int i = 0; i < SPIN; ++i) v = v * 1664525u + 1013904223u;
	if (threadIdx.x == 0) out[blockIdx.x] = v;
}

int main() {
	unsigned *data, *out;
	const unsigned** table;
	(void)hipMalloc(&data, 248 * 64 * 4);
	(void)hipMemset(data, 1, 248 * 64 * 4);
	(void)hipMalloc(&out, 4096 * 4);
	(void)hipMalloc(&table, 248 * sizeof(void*));
	Big big{};
	big.n = 96;
	std::vector<const unsigned*> ptrs(248);
	for (int i = 0; i < 248; ++i) { big.entry[i] = data + i * 64; ptrs[i] = data + i * 64; }
	(void)hipMemcpy(table, ptrs.data(), 248 * sizeof(void*), hipMemcpyHostToDevice);
	Small small{96, 0, table};
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int launches = 2000;
	for (int grid : {256, 768, 3072}) {
		for (int variant = 0; variant < 4; ++variant) {
			float best = 1e9f;
			for (int rep = 0; rep < 3; ++rep) {
				(void)hipEventRecord(e0, 0);
				for (int i = 0; i < launches; ++i) {
					if (variant == 0) hipLaunchKernelGGL(k_big<0>, dim3(grid), dim3(512), 0, 0, big, out);
					if (variant == 1) hipLaunchKernelGGL(k_small<0>, dim3(grid), dim3(512), 0, 0, small, out);
					if (variant == 2) hipLaunchKernelGGL(k_big<2000>, dim3(grid), dim3(512), 0, 0, big, out);
					if (variant == 3) hipLaunchKernelGGL(k_small<2000>, dim3(grid), dim3(512), 0, 0, small, out);
				}
				(void)hipEventRecord(e1, 0);
				(void)hipEventSynchronize(e1);
				float ms = 0;
				(void)hipEventElapsedTime(&ms, e0, e1);
				best = ms < best ? ms : best;
			}
			const char* names[4] = {"2 KB of arguments, trivial kernel", "24 bytes + table in memory, trivial kernel", "2 KB of arguments, ~10 us kernel", "24 bytes + table in memory, ~10 us kernel"};
			printf("grid %4d  %-44s %7.3f us per launch (%d back-to-back launches on one stream)\n", grid, names[variant], best * 1e3 / launches, launches);
		}
	}
	return 0;
}
