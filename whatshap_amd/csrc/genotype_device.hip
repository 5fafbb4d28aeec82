// genotype_device.hip -- device path of GenotypeDPTable (genotype.h; src/genotypedptable.cpp:200-441).
//
// Scaled forward-backward over the columns of the phasing path.  With x a bipartition of column c, i / j transmission
// values and a an allele assignment:
//
//   cost_i(x, a)   = prod over the partitions p of W_i(x)[p][(a >> p) & 1],  W_i(x)[p][al] = prod over the reads r on the side of
//                    x that transmission value i maps to partition p of (al == allele_r ? 1 - e_r : e_r)
//                    (GenotypeColumnCostComputer, src/genotypecolumncostcomputer.cpp:52-103)
//   backward (:200-289)   B_{c-1}[y][j] = sum over x with back(x) = y, over i, a of  B_c[fwd(x)][i] * cost_i(x, a) * P(j -> i) * prior_c(i, a)
//   forward  (:292-441)   alpha_c(x, i, a) = (sum_j A_{c-1}[back(x)][j] * P(j -> i)) * cost_i(x, a) * prior_c(i, a),
//                         A_c[fwd(x)][i] += alpha_c(x, i, a),   L_c[individual][genotype under (i, a)] += alpha_c(x, i, a) * B_c[fwd(x)][i]
//   output               L_c / sum(L_c): any constant factor on a whole A or B column cancels, so columns are rescaled freely
//                        (every column is stored as written plus the per-block sums of what was written; readers multiply by
//                        the reciprocal of the total -- no extra pass, same value in every reader).
//
// One launch per column and direction, one thread per projection entry (the reads that start / end in the column are
// looped over; columns where many do are split further and accumulate with atomics).  Backward columns are kept at the
// end of every WINDOW of columns only and recomputed window by window in front of the forward pass, as the reference does
// with its sqrt(n) checkpoints (:135-159, :313-327).  Arithmetic is f64 (the reference: long double): parity is to a
// tolerance.  This is the first device version of the row: correct and measured, not yet run-fused like the phasing path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "debug_build.h"
#include "device_pool.h"
#include "genotype.h"

#define GENO_TRY(expr)                                                                                   \
	do {                                                                                                 \
		hipError_t e_ = (expr);                                                                          \
		if (e_ != hipSuccess) {                                                                          \
			msg = std::string(#expr) + " failed: " + hipGetErrorString(e_);                              \
			return WHAMD_ERR_DEVICE;                                                                     \
		}                                                                                                \
	} while (0)

namespace whamd {

namespace {

constexpr int GENO_BLOCK = 256;
constexpr int GENO_MAXA = 16;      // allele assignments (P <= 4)
constexpr int GENO_MAXGL = 1 + 3 * MAX_IND;
constexpr uint32_t GENO_LOOP_BITS = 2;   // a thread loops over at most 4 cells of its projection entry

struct GenoDev {
	const uint64_t* col_ptr;
	const uint8_t* ent_ind;
	const uint8_t* ent_allele;
	const double* ent_pe;
	const uint8_t* k;
	const uint8_t* b;
	const uint8_t* f;
	const uint32_t* fwd_mask;
	const double* bern;      // [n_cols][nb]
	const double* prior;     // [n_cols][T][A]
	const uint8_t* gidx;     // [T][A][n_ind]
	const int8_t* h2p;       // [T][n_ind][2]
	uint32_t T, A, P, n_ind, nb, n_cols;
	const double* tables;     // lookup tables of every column, written once per solve by geno_tables (table_stride doubles per column)
	uint32_t table_stride, pad_t;
	// Product slots: founders first, in partition order (slot p = the haplotype that IS partition p), then the two
	// haplotypes of every child.  slot_of[individual * 2 + haplotype]; child_part[i][q] = partition child slot q joins under
	// transmission value i.
	uint32_t n_child_slots;
	uint8_t slot_of[2 * MAX_IND];
	uint8_t child_part[MAX_T][4];
};

// what a block stages in LDS for its column
struct GenoShared {
	double pe[MAX_COVERAGE];
	double prior[MAX_T * GENO_MAXA];
	double bern[5];
	double red[GENO_BLOCK / 64][GENO_MAXGL + MAX_T];
	double inv[2];
	uint8_t ind[MAX_COVERAGE + 7], allele[MAX_COVERAGE + 7];
	uint8_t gidx[MAX_T * GENO_MAXA * MAX_IND];
	int8_t h2p[MAX_T * MAX_IND * 2];
	uint8_t slot_of[2 * MAX_IND + 4];     // (copies of the kernel arguments: indexed per thread, which arguments cannot be)
	uint8_t child_part[MAX_T][4];
};

__device__ __forceinline__ void geno_stage(const GenoDev& G, uint32_t c, uint32_t k, GenoShared& S) {
	const uint32_t tid = threadIdx.x;
	const uint64_t e0 = G.col_ptr[c];
	if (tid < k) { S.pe[tid] = G.ent_pe[e0 + tid]; S.ind[tid] = G.ent_ind[e0 + tid]; S.allele[tid] = G.ent_allele[e0 + tid]; }
	for (uint32_t i = tid; i < G.T * G.A; i += GENO_BLOCK) S.prior[i] = G.prior[(size_t)c * G.T * G.A + i];
	for (uint32_t i = tid; i < G.T * G.A * G.n_ind; i += GENO_BLOCK) S.gidx[i] = G.gidx[i];
	for (uint32_t i = tid; i < G.T * G.n_ind * 2; i += GENO_BLOCK) S.h2p[i] = G.h2p[i];
	if (tid < G.nb) S.bern[tid] = G.bern[(size_t)c * G.nb + tid];
	if (tid == 64) {   // (static indices: the arguments stay in scalar registers)
#pragma unroll
		for (int q = 0; q < 2 * MAX_IND; ++q) S.slot_of[q] = G.slot_of[q];
	}
	if (tid == 128) {
#pragma unroll
		for (int q = 0; q < MAX_T; ++q) {
#pragma unroll
			for (int r = 0; r < 4; ++r) S.child_part[q][r] = G.child_part[q][r];
		}
	}
}

constexpr uint32_t GENO_GROUP_BITS = 7;      // reads per lookup table
constexpr uint32_t GENO_GROUP = 1u << GENO_GROUP_BITS;
constexpr int GENO_MAXSLOTS = 8;             // 2 * individuals (P <= 4: at most a quartet)

// Lookup tables of a column (global memory, geno_tables): for every group of 7 reads and every setting of their bits, the product over the
// group's reads of the emission factors, per product slot and allele:  tab[(group * 128 + bits) * E + slot * 2 + allele],
// E = 4 * individuals.  A read r with bit b belongs to haplotype b ^ 1 of its individual (bit 0 <-> "entry_in_partition1",
// src/genotypecolumncostcomputer.cpp:61) and contributes (allele == allele_r ? 1 - e_r : e_r).
// One launch for ALL columns (blockIdx.y = column): building a column's tables inside its own step kernel cost 2.9 us of
// every 10.7 us dependent launch; here it is a few hundred microseconds of fully parallel work per solve.
__global__ __launch_bounds__(GENO_BLOCK) void geno_tables(GenoDev G, double* __restrict__ tables) {
	const uint32_t c = blockIdx.y, k = G.k[c];
	const uint32_t E = 4u * G.n_ind, groups = (k + GENO_GROUP_BITS - 1u) / GENO_GROUP_BITS;
	const uint32_t idx = blockIdx.x * GENO_BLOCK + threadIdx.x;
	if (idx >= groups * GENO_GROUP) return;
	const uint64_t e0 = G.col_ptr[c];
	double v[GENO_MAXSLOTS][2];   // the entry in registers: static indices, the slot of a read is matched with selects
#pragma unroll
	for (int q = 0; q < GENO_MAXSLOTS; ++q) v[q][0] = v[q][1] = 1.0;
	const uint32_t g = idx >> GENO_GROUP_BITS, bits = idx & (GENO_GROUP - 1u);
#pragma unroll
	for (uint32_t jj = 0; jj < GENO_GROUP_BITS; ++jj) {
		const uint32_t j = g * GENO_GROUP_BITS + jj;
		if (j >= k) break;
		const uint32_t al = G.ent_allele[e0 + j];
		if (al > 1u) continue;   // BLANK
		const uint32_t key = (uint32_t)G.ent_ind[e0 + j] * 2u + (((bits >> jj) & 1u) ^ 1u);   // individual * 2 + haplotype
		uint32_t slot = 0;
#pragma unroll
		for (int q = 0; q < 2 * MAX_IND; ++q) slot = key == (uint32_t)q ? (uint32_t)G.slot_of[q] : slot;
		const double pe = G.ent_pe[e0 + j], ok = 1.0 - pe;
		const double m0 = al == 0u ? ok : pe, m1 = al == 0u ? pe : ok;
#pragma unroll
		for (int q = 0; q < GENO_MAXSLOTS; ++q) {
			v[q][0] *= slot == (uint32_t)q ? m0 : 1.0;
			v[q][1] *= slot == (uint32_t)q ? m1 : 1.0;
		}
	}
	double* e = tables + (size_t)c * G.table_stride + (size_t)idx * E;
#pragma unroll
	for (int q = 0; q < GENO_MAXSLOTS; ++q)
		if ((uint32_t)q * 2u < E) *reinterpret_cast<double2*>(e + q * 2) = make_double2(v[q][0], v[q][1]);
}

// The column's tables, global -> LDS: one coalesced copy in the same barrier phase as the staging of the column's reads
__device__ __forceinline__ void geno_copy_tables(const GenoDev& G, uint32_t c, uint32_t k, double* lds) {
	const uint32_t E = 4u * G.n_ind, groups = (k + GENO_GROUP_BITS - 1u) / GENO_GROUP_BITS;
	const double2* __restrict__ src = reinterpret_cast<const double2*>(G.tables + (size_t)c * G.table_stride);
	double2* dst = reinterpret_cast<double2*>(lds);
	for (uint32_t i = threadIdx.x; i < groups * GENO_GROUP * E / 2u; i += GENO_BLOCK) dst[i] = src[i];
}

// V[slot][allele] of one cell: the product of its groups' table entries
__device__ __forceinline__ void geno_cell_products(const GenoDev& G, uint32_t k, uint32_t x, const double* tab, double (&V)[GENO_MAXSLOTS][2]) {
	const uint32_t E = 4u * G.n_ind, groups = (k + GENO_GROUP_BITS - 1u) / GENO_GROUP_BITS;
#pragma unroll
	for (int q = 0; q < GENO_MAXSLOTS; ++q) V[q][0] = V[q][1] = 1.0;
	for (uint32_t g = 0; g < groups; ++g) {
		const double* e = tab + ((size_t)g * GENO_GROUP + ((x >> (g * GENO_GROUP_BITS)) & (GENO_GROUP - 1u))) * E;
#pragma unroll
		for (int q = 0; q < GENO_MAXSLOTS; ++q) {
			if ((uint32_t)q * 2u < E) {
				const double2 v = *reinterpret_cast<const double2*>(e + q * 2);
				V[q][0] *= v.x;
				V[q][1] *= v.y;
			}
		}
	}
}

// W_i[p][allele]: the founder haplotype that is partition p times the child haplotypes that join it under transmission value i
__device__ __forceinline__ void geno_partition_products(const GenoDev& G, const GenoShared& S, const double (&V)[GENO_MAXSLOTS][2], uint32_t i, double (&W)[4][2]) {
#pragma unroll
	for (int p = 0; p < 4; ++p) { W[p][0] = V[p][0]; W[p][1] = V[p][1]; }   // (slots >= P hold children or ones)
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		if ((uint32_t)q < G.n_child_slots) {
			const uint32_t part = S.child_part[i][q];
			const double c0 = V[4 + q][0], c1 = V[4 + q][1];   // children exist only below two founders: P == 4, child slot q is V[4 + q]
#pragma unroll
			for (int p = 0; p < 4; ++p) {
				W[p][0] *= part == (uint32_t)p ? c0 : 1.0;
				W[p][1] *= part == (uint32_t)p ? c1 : 1.0;
			}
		}
	}
}

__device__ __forceinline__ double geno_assignment_cost(const double (&W)[4][2], uint32_t P, uint32_t a) {
	double cst = 1.0;
#pragma unroll
	for (int p = 0; p < 4; ++p)
		if ((uint32_t)p < P) cst *= W[p][(a >> p) & 1u];
	return cst;
}

__device__ __forceinline__ uint32_t geno_pext(uint32_t x, uint32_t mask) {
	uint32_t r = 0, o = 0;
	while (mask) {
		const uint32_t low = mask & (0u - mask);
		r |= ((x & low) ? 1u : 0u) << o++;
		mask ^= low;
	}
	return r;
}
__device__ __forceinline__ uint32_t geno_pdep(uint32_t v, uint32_t mask) {
	uint32_t r = 0;
	while (mask) {
		const uint32_t low = mask & (0u - mask);
		if (v & 1u) r |= low;
		v >>= 1;
		mask ^= low;
	}
	return r;
}

// per-column parameters the host knows: passed by value, nothing on the kernels' critical path loads them
struct GenoCol {
	uint32_t c, k, b, f, fmask, loop_bits, use_atomics, pad;
};

// the per-thread sum of the partials of a stored column (issued early; reduced by geno_inverse_finish)
__device__ __forceinline__ double geno_partials_begin(const double* partials, uint32_t n_blocks) {
	double v = 0.0;
	for (uint32_t i = threadIdx.x; i < n_blocks; i += GENO_BLOCK) v += partials[i];
	return v;
}
__device__ __forceinline__ double geno_inverse_finish(double v, double* red) {
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
	if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	double total = 0.0;
	for (int w = 0; w < GENO_BLOCK / 64; ++w) total += red[w];
	__syncthreads();
	return total > 0.0 ? 1.0 / total : 0.0;
}

// Thread mapping of both column kernels: thread = (projection entry, transmission value i); the T threads of an entry are
// neighbouring lanes.  A thread loops over at most 4 cells of its entry (the reads that start / end in the column).

// Backward step of column c: reads B_c (`in`, null for the last column), writes B_{c-1} (`out`, 2^b_c x T) and the per-block sums.
template <int T>
__global__ __launch_bounds__(GENO_BLOCK) void geno_backward(GenoDev G, GenoCol C, const double* __restrict__ in, const double* __restrict__ in_partials,
                                                           uint32_t in_blocks, double* __restrict__ out, double* __restrict__ out_partials) {
	__shared__ GenoShared S;
	extern __shared__ __attribute__((aligned(16))) double geno_tab[];
	const uint32_t k = C.k, b = C.b, fmask = C.fmask, loop_bits = C.loop_bits;
	// every global load of the kernel is issued here, before the first barrier: one memory round trip on the critical path
	// (the column's scale -- the reciprocal of the sum of its partials -- is linear in the result and applied at the end)
	const double psum = in ? geno_partials_begin(in_partials, in_blocks) : 0.0;
	const uint64_t n_entries = 1ull << (k - loop_bits);
	const uint64_t t = (uint64_t)blockIdx.x * GENO_BLOCK + threadIdx.x;
	const uint64_t entry = t / T;
	const uint32_t i = (uint32_t)(t % T);
	const bool active = entry < n_entries;
	const uint32_t y = (uint32_t)entry & ((1u << b) - 1u), chunk = (uint32_t)(entry >> b);
	double beta_raw[1u << GENO_LOOP_BITS];
#pragma unroll
	for (uint32_t e = 0; e < (1u << GENO_LOOP_BITS); ++e) {
		beta_raw[e] = 1.0;
		if (in && active && e < (1u << loop_bits)) beta_raw[e] = in[(size_t)geno_pext(y | (((chunk << loop_bits) | e) << b), fmask) * T + i];
	}
	geno_copy_tables(G, C.c, k, geno_tab);
	geno_stage(G, C.c, k, S);
	__syncthreads();
	double partial = 0.0;   // sum over this thread's cells of beta * sum_a prior * cost, for its transmission value i
	if (active) {
#pragma unroll
		for (uint32_t e = 0; e < (1u << GENO_LOOP_BITS); ++e) {
			if (e >= (1u << loop_bits)) break;
			const double beta = beta_raw[e];
			double V[GENO_MAXSLOTS][2], W[4][2];
			geno_cell_products(G, k, y | (((chunk << loop_bits) | e) << b), geno_tab, V);
			geno_partition_products(G, S, V, i, W);
			double s = 0.0;
			for (uint32_t a = 0; a < G.A; ++a) s += S.prior[i * G.A + a] * geno_assignment_cost(W, G.P, a);
			partial += s * beta;
		}
	}
	// out[y][j] = sum_i partial_i * P(j -> i): thread i of the entry produces j = i from its neighbours' partials
	double acc = 0.0;
	const uint32_t lane = threadIdx.x & 63u, base = lane & ~(uint32_t)(T - 1);
#pragma unroll
	for (int ii = 0; ii < T; ++ii) {
		const double other = T == 1 ? partial : __shfl(partial, (int)(base + ii));
		acc += other * S.bern[__popc((uint32_t)ii ^ i)];
	}
	const double inv = in ? geno_inverse_finish(psum, &S.red[0][0]) : 1.0;
	acc *= inv;
	if (active) {
		if (C.use_atomics) atomicAdd(out + (size_t)y * T + i, acc);
		else out[(size_t)y * T + i] = acc;
	} else acc = 0.0;
	double v = acc;
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
	__syncthreads();
	if ((threadIdx.x & 63u) == 0) S.red[threadIdx.x >> 6][0] = v;
	__syncthreads();
	if (threadIdx.x == 0) {
		double total = 0.0;
		for (int w = 0; w < GENO_BLOCK / 64; ++w) total += S.red[w][0];
		out_partials[blockIdx.x] = total;
	}
}

// Forward step of column c: reads A_{c-1} (`prev`, null for column 0) and B_c (`beta`, null for the last column), writes A_c
// (`out`, null for the last column) with its per-block sums, and the per-block sums of the normalisation and of the genotype
// likelihood numerators (gl_partials[block][1 + 3 * individuals]).
// MODE 0: all of it (windowed solve).  When every column fits in memory the forward chain runs BESIDE the backward chain
// instead of after it (two streams, half as many dependent launches): MODE 1 writes A_c only, and MODE 2 -- one launch for
// a batch of columns, blockIdx.y = column, arguments from a device array -- combines the stored A_{c-1} and B_c into the
// likelihood sums (every column is independent there).
struct GenoFwdArgs {
	GenoCol C;
	const double* prev; const double* prev_partials;
	const double* beta; const double* beta_partials;
	double* out; double* out_partials;
	double* gl_partials;
	uint32_t prev_blocks, beta_blocks, n_blocks, pad;
};

template <int T, int MODE>
__global__ __launch_bounds__(GENO_BLOCK) void geno_forward(GenoDev G, GenoFwdArgs by_value, const GenoFwdArgs* __restrict__ batch) {
	__shared__ GenoShared S;
	extern __shared__ __attribute__((aligned(16))) double geno_tab[];
	const GenoFwdArgs A = MODE == 2 ? batch[blockIdx.y] : by_value;
	if (MODE == 2 && blockIdx.x >= A.n_blocks) return;
	const GenoCol C = A.C;
	const double* __restrict__ prev = A.prev;
	const double* __restrict__ prev_partials = A.prev_partials;
	const double* __restrict__ beta = MODE == 1 ? nullptr : A.beta;
	const double* __restrict__ beta_partials = A.beta_partials;
	double* __restrict__ out = MODE == 2 ? nullptr : A.out;
	double* __restrict__ out_partials = A.out_partials;
	double* __restrict__ gl_partials = A.gl_partials;
	const uint32_t prev_blocks = A.prev_blocks, beta_blocks = A.beta_blocks;
	const uint32_t k = C.k, b = C.b, f = C.f, fmask = C.fmask, loop_bits = C.loop_bits;   // (last column: f = 0, fmask = 0)
	const double psum_prev = prev ? geno_partials_begin(prev_partials, prev_blocks) : 0.0;
	const double psum_beta = beta ? geno_partials_begin(beta_partials, beta_blocks) : 0.0;
	const uint32_t kmask = k >= 32u ? 0xFFFFFFFFu : ((1u << k) - 1u), endmask = kmask & ~fmask;
	const uint64_t n_entries = 1ull << (k - loop_bits);
	const uint64_t t = (uint64_t)blockIdx.x * GENO_BLOCK + threadIdx.x;
	const uint64_t entry = t / T;
	const uint32_t i = (uint32_t)(t % T);
	const uint32_t n_gl = 1u + 3u * G.n_ind;
	// all global loads up front (see geno_backward); the two scales are applied at the end
	const bool active = entry < n_entries;
	const uint32_t yf0 = (uint32_t)entry & ((1u << f) - 1u), chunk0 = (uint32_t)(entry >> f);
	const uint32_t xf0 = geno_pdep(yf0, fmask);
	const double bt_raw = (beta && active) ? beta[(size_t)yf0 * T + i] : 1.0;
	double prev_raw[1u << GENO_LOOP_BITS][T];
#pragma unroll
	for (uint32_t e = 0; e < (1u << GENO_LOOP_BITS); ++e) {
		const uint32_t x = xf0 | geno_pdep((chunk0 << loop_bits) | e, endmask);
		const bool use = prev && active && e < (1u << loop_bits);
#pragma unroll
		for (int j = 0; j < T; ++j) prev_raw[e][j] = use ? prev[(size_t)(x & ((1u << b) - 1u)) * T + j] : 0.0;
	}
	geno_copy_tables(G, C.c, k, geno_tab);
	geno_stage(G, C.c, k, S);
	__syncthreads();
	double fa[GENO_MAXA];   // per allele assignment: sum over this thread's cells of forward * backward (its transmission value)
#pragma unroll
	for (int a = 0; a < GENO_MAXA; ++a) fa[a] = 0.0;
	double acc = 0.0;
	if (active) {
		const double bt = bt_raw;
#pragma unroll
		for (uint32_t e = 0; e < (1u << GENO_LOOP_BITS); ++e) {
			if (e >= (1u << loop_bits)) break;
			const uint32_t x = xf0 | geno_pdep((chunk0 << loop_bits) | e, endmask);
			double sum_prev = 1.0;
			if (prev) {
				sum_prev = 0.0;
#pragma unroll
				for (int j = 0; j < T; ++j) sum_prev += prev_raw[e][j] * S.bern[__popc((uint32_t)j ^ i)];
			}
			double V[GENO_MAXSLOTS][2], W[4][2];
			geno_cell_products(G, k, x, geno_tab, V);
			geno_partition_products(G, S, V, i, W);
#pragma unroll
			for (int a = 0; a < GENO_MAXA; ++a) {
				if ((uint32_t)a < G.A) {
					const double fw = sum_prev * geno_assignment_cost(W, G.P, (uint32_t)a) * S.prior[i * G.A + a];
					acc += fw;
					if (MODE != 1) fa[a] += fw * bt;
				}
			}
		}
	}
	{
		const double inv_prev = prev ? geno_inverse_finish(psum_prev, &S.red[0][0]) : 1.0;
		const double inv_beta = beta ? geno_inverse_finish(psum_beta, &S.red[0][0]) : 1.0;
		acc *= inv_prev;
		const double both = inv_prev * inv_beta;
#pragma unroll
		for (int a = 0; a < GENO_MAXA; ++a) fa[a] *= both;
		if (active && out) {
			if (C.use_atomics) atomicAdd(out + (size_t)yf0 * T + i, acc);
			else out[(size_t)yf0 * T + i] = acc;
		}
	}
	if (MODE == 1) {   // A_c only: the per-block sum of what was written
		double v = acc;
		for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
		__syncthreads();
		if ((threadIdx.x & 63u) == 0) S.red[threadIdx.x >> 6][0] = v;
		__syncthreads();
		if (threadIdx.x == 0) {
			double total = 0.0;
			for (int w = 0; w < GENO_BLOCK / 64; ++w) total += S.red[w][0];
			out_partials[blockIdx.x] = total;
		}
		return;
	}
	// marginalise this thread's assignments over the genotypes (src/genotypedptable.cpp:376-383), once per thread
	double gl[GENO_MAXGL];
#pragma unroll
	for (int q = 0; q < GENO_MAXGL; ++q) gl[q] = 0.0;
#pragma unroll
	for (int a = 0; a < GENO_MAXA; ++a) {
		if ((uint32_t)a < G.A) {
			gl[0] += fa[a];
			const uint8_t* gi = S.gidx + ((size_t)i * G.A + a) * G.n_ind;
#pragma unroll
			for (int s = 0; s < 4; ++s) {
				if ((uint32_t)s < G.n_ind) {
					const uint32_t g = gi[s];
					gl[1 + 3 * s + 0] += g == 0u ? fa[a] : 0.0;
					gl[1 + 3 * s + 1] += g == 1u ? fa[a] : 0.0;
					gl[1 + 3 * s + 2] += g == 2u ? fa[a] : 0.0;
				}
			}
		}
	}
	// per-block sums: the likelihood numerators (slots 0 .. n_gl - 1) and the written column (slot n_gl)
	__syncthreads();
#pragma unroll
	for (int q = 0; q <= 13; ++q) {   // 1 + 3 * 4 individuals + the column sum
		if ((uint32_t)q <= n_gl) {    // (wave-uniform)
			double v = (uint32_t)q == n_gl ? acc : gl[q < GENO_MAXGL ? q : 0];
			for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
			if ((threadIdx.x & 63u) == 0) S.red[threadIdx.x >> 6][q] = v;
		}
	}
	__syncthreads();
	if (threadIdx.x <= n_gl) {
		double total = 0.0;
		for (int w = 0; w < GENO_BLOCK / 64; ++w) total += S.red[w][threadIdx.x];
		if (threadIdx.x == n_gl) { if (out) out_partials[blockIdx.x] = total; }
		else gl_partials[(size_t)blockIdx.x * n_gl + threadIdx.x] = total;
	}
}

// Normalised genotype likelihoods of the columns of one window: block = column.
__global__ __launch_bounds__(64) void geno_finish(const double* __restrict__ gl_partials, const uint32_t* __restrict__ fw_blocks, uint32_t c0,
                                                  uint32_t max_blocks, uint32_t n_ind, uint32_t n_cols, double* __restrict__ gl_out) {
	const uint32_t ci = blockIdx.x, c = c0 + ci, n_gl = 1u + 3u * n_ind, nb = fw_blocks[c];
	__shared__ double tot[GENO_MAXGL];
	const double* p = gl_partials + (size_t)ci * max_blocks * n_gl;
	if (threadIdx.x < n_gl) {
		double v = 0.0;
		for (uint32_t blk = 0; blk < nb; ++blk) v += p[(size_t)blk * n_gl + threadIdx.x];
		tot[threadIdx.x] = v;
	}
	__syncthreads();
	if (threadIdx.x >= 1 && threadIdx.x < n_gl) {
		const uint32_t s = (threadIdx.x - 1) / 3, g = (threadIdx.x - 1) % 3;
		gl_out[((size_t)s * n_cols + c) * 3 + g] = tot[threadIdx.x] / tot[0];
	}
}

uint32_t blocks_for(uint32_t k, uint32_t proj, uint32_t T) {   // grid of a column kernel: 2^(k - min(k - proj, LOOP)) entries x T threads
	const uint32_t nfree = k - proj, loop_bits = std::min(nfree, GENO_LOOP_BITS);
	const uint64_t threads = (1ull << (k - loop_bits)) * T;
	return (uint32_t)((threads + GENO_BLOCK - 1) / GENO_BLOCK);
}

}  // namespace

// The column store (tens of GB) is kept between calls, one block per device: mapping that much fresh device memory took 3 - 4 s
// in about one call out of four (hipMalloc right after the hipFree of the previous call), reusing the block costs nothing.
// A call that finds the block in use (another thread) or too small allocates its own.  genotype_release_cache() frees them.
namespace {
struct SlabCache { void* ptr = nullptr; size_t bytes = 0; bool in_use = false; };
SlabCache g_slab[16];
std::mutex g_slab_mutex;
}  // namespace

void* genotype_slab_acquire(int device, size_t bytes) {
	if (device < 0 || device >= 16 || debug_env("WHAMD_GENOTYPE_NO_CACHE")) return nullptr;
	std::lock_guard<std::mutex> lock(g_slab_mutex);
	SlabCache& sc = g_slab[device];
	if (sc.in_use) return nullptr;
	if (sc.bytes < bytes) {
		if (sc.ptr) (void)hipFree(sc.ptr);
		sc = SlabCache();
		hipError_t e = hipMalloc(&sc.ptr, bytes);
		if (e != hipSuccess) { (void)hipGetLastError(); devpool_release(); e = hipMalloc(&sc.ptr, bytes); }
		if (e == hipSuccess) sc.bytes = bytes;
		else { sc = SlabCache(); (void)hipGetLastError(); return nullptr; }
	}
	sc.in_use = true;
	return sc.ptr;
}

void genotype_slab_release(int device) {
	if (device < 0 || device >= 16) return;
	std::lock_guard<std::mutex> lock(g_slab_mutex);
	SlabCache& sc = g_slab[device];
	sc.in_use = false;
	size_t free_b = 0, total_b = 0;
	if (sc.ptr && hipMemGetInfo(&free_b, &total_b) == hipSuccess && sc.bytes > total_b / 4) {
		(void)hipFree(sc.ptr);
		sc = SlabCache();
	}
}

size_t genotype_slab_idle_bytes(int device) {
	if (device < 0 || device >= 16) return 0;
	std::lock_guard<std::mutex> lock(g_slab_mutex);
	return g_slab[device].in_use ? 0 : g_slab[device].bytes;
}

void genotype_release_cache() {
	std::lock_guard<std::mutex> lock(g_slab_mutex);
	int current = 0;
	(void)hipGetDevice(&current);
	for (int d = 0; d < 16; ++d) {
		if (g_slab[d].ptr && !g_slab[d].in_use) {
			(void)hipSetDevice(d);
			(void)hipFree(g_slab[d].ptr);
			g_slab[d] = SlabCache();
		}
	}
	(void)hipSetDevice(current);
}

whamd_status_t genotype_solve_device(const Problem& p, const GenotypeModel& m, int device, uint32_t window_hint,
                                     std::vector<double>& gl_out, GenotypeStats& st, std::string& msg) {
	const uint32_t n = p.n_cols, T = p.T, ni = p.n_ind;
	gl_out.assign((size_t)ni * n * 3, 0.0);
	st = GenotypeStats();
	st.n_columns = n;
	st.transmissions = T;
	if (n == 0) return WHAMD_OK;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		msg = "no HIP device visible: the whatshap_amd device path needs an MI355X (gfx950); there is no CPU fallback";
		return WHAMD_ERR_DEVICE;
	}
	if (device < 0 || device >= ndev) {
		msg = "device index " + std::to_string(device) + " out of range (" + std::to_string(ndev) + " visible)";
		return WHAMD_ERR_DEVICE;
	}
	if (T != 1 && T != 4 && T != 16) { msg = "unsupported number of transmission values"; return WHAMD_ERR_UNSUPPORTED; }
	GENO_TRY(hipSetDevice(device));
	// the run-fused path wherever it applies (no forced window, every column in a run, stores fit in HBM)
	if (!window_hint && !debug_env("WHAMD_GENOTYPE_COLUMNS")) {
		bool used = false;
		const whamd_status_t sst = genotype_solve_slots(p, m, device, gl_out, st, used, msg);
		if (sst != WHAMD_OK) return sst;
		if (used) return WHAMD_OK;
		gl_out.assign((size_t)ni * n * 3, 0.0);
		st = GenotypeStats();
		st.n_columns = n;
		st.transmissions = T;
	}
	uint32_t max_k = 0, max_proj = 0;
	for (uint32_t c = 0; c < n; ++c) {
		max_k = std::max<uint32_t>(max_k, p.k[c]);
		max_proj = std::max<uint32_t>(max_proj, std::max<uint32_t>(p.f[c], p.b[c]));
		st.n_cells += 1ull << p.k[c];
	}
	st.max_coverage = max_k;
	const size_t buf_doubles = ((size_t)1 << max_proj) * T;
	const uint32_t max_blocks = (uint32_t)(((((size_t)1 << max_k) * T) + GENO_BLOCK - 1) / GENO_BLOCK);
	const uint32_t n_gl = 1 + 3 * ni;
	size_t free_b = 0, total_b = 0;
	GENO_TRY(hipMemGetInfo(&free_b, &total_b));
	if (free_b < total_b / 2) {   // a phasing table of this process may have left its arena in the cache (dp_device.hip)
		dptable_release_arena_cache();
		GENO_TRY(hipMemGetInfo(&free_b, &total_b));
	}
	free_b += genotype_slab_idle_bytes(device);   // the block kept from an earlier call is available to this one
	// Window = how many backward columns are kept at once.  If all of them fit in a quarter of the free memory there is one
	// window and no column is computed twice; otherwise the reference's scheme: sqrt(n) kept columns, the rest recomputed.
	uint32_t K = window_hint;
	if (!K) {
		const double per_column = (double)buf_doubles * 8 + (double)max_blocks * 8 * (1 + n_gl);
		K = 2.0 * per_column * n <= 0.4 * (double)free_b ? n : (uint32_t)std::ceil(std::sqrt((double)n));   // (backward AND forward columns kept)
	}
	K = std::max(1u, std::min(K, n));
	st.window = K;
	const uint32_t n_windows = (n + K - 1) / K;
	{
		const double need = (double)(buf_doubles * 8 + (size_t)max_blocks * 8) * (n_windows + 2.0 * K + 4.0) + (double)K * max_blocks * n_gl * 8 + (double)p.entries.size() * 10 + (double)n * (64 + 8.0 * T * m.A)
		                    + (double)n * ((max_k + GENO_GROUP_BITS - 1) / GENO_GROUP_BITS) * GENO_GROUP * 4 * ni * 8.0;
		if (need + (double)(1ull << 30) > (double)free_b) {
			msg = "genotyping buffers of " + std::to_string((uint64_t)(need / 1048576.0)) + " MiB do not fit in free HBM";
			return WHAMD_ERR_UNSUPPORTED;
		}
	}
	const auto t_phase0 = std::chrono::steady_clock::now();
	auto phase_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_phase0).count(); };
	hipStream_t stream = nullptr;
	GENO_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
	std::vector<void*> allocations;
	std::vector<hipStream_t> extra_streams;   // the second chain's stream and every event: owned by cleanup(), whichever way the call ends
	std::vector<hipEvent_t> all_events;
	bool slab_from_cache = false;
	auto cleanup = [&]() {
		for (hipEvent_t e : all_events) (void)hipEventDestroy(e);
		for (hipStream_t s2 : extra_streams) (void)hipStreamDestroy(s2);
		for (void* a : allocations) (void)hipFree(a);
		if (slab_from_cache) genotype_slab_release(device);
		if (stream) (void)hipStreamDestroy(stream);
	};
	auto fail = [&](hipError_t e, const char* what) {
		msg = std::string(what) + " failed: " + hipGetErrorString(e);
		cleanup();
		return WHAMD_ERR_DEVICE;
	};
#define GENO_DEV(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_, #expr); } while (0)
	auto alloc = [&](void** dptr, size_t bytes) -> hipError_t {
		hipError_t e = hipMalloc(dptr, std::max<size_t>(bytes, 16));
		if (e != hipSuccess) {   // idle blocks of the phasing / heuristic pool may hold the memory: give them back, try once more
			(void)hipGetLastError();
			devpool_release();
			e = hipMalloc(dptr, std::max<size_t>(bytes, 16));
		}
		if (e == hipSuccess) allocations.push_back(*dptr);
		return e;
	};
	auto up = [&](void** dptr, const void* src, size_t bytes) -> hipError_t {
		hipError_t e = alloc(dptr, bytes);
		if (e == hipSuccess && bytes) e = hipMemcpyAsync(*dptr, src, bytes, hipMemcpyHostToDevice, stream);
		return e;
	};
	// ---- upload the model
	std::vector<uint8_t> ent_ind(p.entries.size()), ent_allele(p.entries.size());
	for (size_t e = 0; e < p.entries.size(); ++e) { ent_ind[e] = p.entries[e].sample; ent_allele[e] = p.entries[e].allele; }
	std::vector<uint32_t> fw_blocks(n), bw_blocks(n);
	for (uint32_t c = 0; c < n; ++c) {
		fw_blocks[c] = blocks_for(p.k[c], c + 1 < n ? p.f[c] : 0u, T);
		bw_blocks[c] = blocks_for(p.k[c], p.b[c], T);
	}
	GenoDev G{};
	void *d_col_ptr, *d_ind, *d_allele, *d_pe, *d_k, *d_b, *d_f, *d_fmask, *d_bern, *d_prior, *d_gidx, *d_h2p, *d_fwb;
	GENO_DEV(up(&d_col_ptr, p.col_ptr.data(), p.col_ptr.size() * 8));
	GENO_DEV(up(&d_ind, ent_ind.data(), ent_ind.size()));
	GENO_DEV(up(&d_allele, ent_allele.data(), ent_allele.size()));
	GENO_DEV(up(&d_pe, m.error_prob.data(), m.error_prob.size() * 8));
	GENO_DEV(up(&d_k, p.k.data(), n));
	GENO_DEV(up(&d_b, p.b.data(), n));
	GENO_DEV(up(&d_f, p.f.data(), n));
	GENO_DEV(up(&d_fmask, p.fwd_mask.data(), (size_t)n * 4));
	GENO_DEV(up(&d_bern, m.transition_bern.data(), m.transition_bern.size() * 8));
	GENO_DEV(up(&d_prior, m.allele_prior.data(), m.allele_prior.size() * 8));
	GENO_DEV(up(&d_gidx, m.genotype_index.data(), m.genotype_index.size()));
	GENO_DEV(up(&d_h2p, p.h2p.data(), p.h2p.size()));
	GENO_DEV(up(&d_fwb, fw_blocks.data(), (size_t)n * 4));
	G.col_ptr = (const uint64_t*)d_col_ptr; G.ent_ind = (const uint8_t*)d_ind; G.ent_allele = (const uint8_t*)d_allele; G.ent_pe = (const double*)d_pe;
	G.k = (const uint8_t*)d_k; G.b = (const uint8_t*)d_b; G.f = (const uint8_t*)d_f; G.fwd_mask = (const uint32_t*)d_fmask;
	G.bern = (const double*)d_bern; G.prior = (const double*)d_prior; G.gidx = (const uint8_t*)d_gidx; G.h2p = (const int8_t*)d_h2p;
	G.T = T; G.A = m.A; G.P = p.P; G.n_ind = ni; G.nb = 2 * p.n_triples + 1; G.n_cols = n;
	{
		// founders are the individuals whose two haplotypes ARE partitions (h2p does not depend on the transmission value)
		std::vector<uint8_t> is_child(ni, 0);
		for (uint32_t t3 = 0; t3 < p.n_triples; ++t3) is_child[p.triples[t3][2]] = 1;
		uint32_t child_slots = 0;
		for (uint32_t s = 0; s < ni; ++s) {
			if (!is_child[s]) {
				G.slot_of[2 * s] = (uint8_t)p.h2p[(size_t)s * 2];
				G.slot_of[2 * s + 1] = (uint8_t)p.h2p[(size_t)s * 2 + 1];
			} else {
				if (p.P != 4 || child_slots + 2 > 4) { cleanup(); msg = "unsupported pedigree shape for device genotyping"; return WHAMD_ERR_UNSUPPORTED; }
				for (uint32_t h = 0; h < 2; ++h) {
					G.slot_of[2 * s + h] = (uint8_t)(p.P + child_slots + h);
					for (uint32_t i = 0; i < T; ++i) G.child_part[i][child_slots + h] = (uint8_t)p.h2p[((size_t)i * ni + s) * 2 + h];
				}
				child_slots += 2;
			}
		}
		G.n_child_slots = child_slots;
	}
	// lookup tables of all columns (geno_tables): one launch before the chains start
	const uint32_t max_groups = (max_k + GENO_GROUP_BITS - 1) / GENO_GROUP_BITS;
	G.table_stride = std::max(1u, max_groups) * GENO_GROUP * 4 * ni;
	double* d_tables = nullptr;
	GENO_DEV(alloc((void**)&d_tables, (size_t)n * G.table_stride * sizeof(double)));
	G.tables = d_tables;
	const size_t table_bytes = (size_t)G.table_stride * sizeof(double);   // the step kernels copy their column's tables into LDS
	if (table_bytes + sizeof(GenoShared) > 160 * 1024) {   // (a quartet beyond coverage ~35: never reached below the 25-read limit, but never a bare launch failure)
		cleanup();
		msg = "lookup tables of " + std::to_string(table_bytes >> 10) + " KiB per column do not fit in LDS";
		return WHAMD_ERR_UNSUPPORTED;
	}
	if (table_bytes + sizeof(GenoShared) > 64 * 1024) {   // more than 64 KiB of LDS needs the opt-in on every kernel that asks for it
		const void* fns[] = {(const void*)geno_backward<1>, (const void*)geno_backward<4>, (const void*)geno_backward<16>,
		                     (const void*)geno_forward<1, 0>, (const void*)geno_forward<1, 1>, (const void*)geno_forward<1, 2>,
		                     (const void*)geno_forward<4, 0>, (const void*)geno_forward<4, 1>, (const void*)geno_forward<4, 2>,
		                     (const void*)geno_forward<16, 0>, (const void*)geno_forward<16, 1>, (const void*)geno_forward<16, 2>};
		for (const void* fn : fns) GENO_DEV(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)sizeof(GenoShared)));
	}
	// ---- buffers: every column buffer carries its per-block sums
	struct Buf { double* v = nullptr; double* partials = nullptr; uint32_t blocks = 0; };
	Buf alpha[2], pp[2];
	std::vector<Buf> ckpt(n_windows), wstore(n_windows == 1 ? K : 2 * (size_t)K), astore(n_windows == 1 ? K : 0);   // windowed: two halves, recompute of window w + 1 beside the forward pass of window w   // astore: the forward columns of the two-chain mode
	{
		// one slab for all of them (tens of thousands of hipMalloc calls would take seconds)
		const size_t count = 4 + (size_t)n_windows + wstore.size() + astore.size();
		double *slab_v = nullptr, *slab_p = nullptr;
		slab_v = (double*)genotype_slab_acquire(device, count * buf_doubles * 8);
		slab_from_cache = slab_v != nullptr;
		if (!slab_v) GENO_DEV(alloc((void**)&slab_v, count * buf_doubles * 8));
		GENO_DEV(alloc((void**)&slab_p, count * (size_t)max_blocks * 8));
		size_t next = 0;
		auto take = [&](Buf& bf) { bf.v = slab_v + next * buf_doubles; bf.partials = slab_p + next * max_blocks; ++next; };
		for (Buf& bf : alpha) take(bf);
		for (Buf& bf : pp) take(bf);
		for (Buf& bf : ckpt) take(bf);
		for (Buf& bf : wstore) take(bf);
		for (Buf& bf : astore) take(bf);
	}
	double *d_glpart = nullptr, *d_gl = nullptr;
	GENO_DEV(alloc((void**)&d_glpart, (size_t)std::min<uint32_t>(K, n_windows == 1 ? 1024u : K) * max_blocks * n_gl * 8));
	GENO_DEV(alloc((void**)&d_gl, gl_out.size() * 8));
	hipEvent_t ev[3];
	for (hipEvent_t& e : ev) { GENO_DEV(hipEventCreate(&e)); all_events.push_back(e); }
	uint64_t launches = 0;
	// one backward step: column c, B_c in `in` (null: last column) -> B_{c-1} in `out`
	auto backward = [&](uint32_t c, const Buf* in, Buf& out, hipStream_t on = nullptr) -> hipError_t {
		if (!on) on = stream;
		const uint32_t blocks = bw_blocks[c];
		const uint32_t atomics = (uint32_t)p.k[c] - p.b[c] > GENO_LOOP_BITS ? 1u : 0u;
		const GenoCol C{c, p.k[c], p.b[c], p.f[c], p.fwd_mask[c], std::min<uint32_t>((uint32_t)p.k[c] - p.b[c], GENO_LOOP_BITS), atomics, 0u};
		if (atomics) { hipError_t e = hipMemsetAsync(out.v, 0, ((size_t)T << p.b[c]) * 8, on); if (e != hipSuccess) return e; }
		out.blocks = blocks;
		const double* iv = in ? in->v : nullptr;
		const double* ip = in ? in->partials : nullptr;
		const uint32_t ib = in ? in->blocks : 0u;
		if (T == 1) hipLaunchKernelGGL(geno_backward<1>, dim3(blocks), dim3(GENO_BLOCK), table_bytes, on, G, C, iv, ip, ib, out.v, out.partials);
		else if (T == 4) hipLaunchKernelGGL(geno_backward<4>, dim3(blocks), dim3(GENO_BLOCK), table_bytes, on, G, C, iv, ip, ib, out.v, out.partials);
		else hipLaunchKernelGGL(geno_backward<16>, dim3(blocks), dim3(GENO_BLOCK), table_bytes, on, G, C, iv, ip, ib, out.v, out.partials);
		++launches;
		return hipGetLastError();
	};
	const double ms_allocated = phase_ms();
	const auto t_enqueue0 = std::chrono::steady_clock::now();
	GENO_DEV(hipEventRecord(ev[0], stream));
	if (max_groups) {
		hipLaunchKernelGGL(geno_tables, dim3((max_groups * GENO_GROUP + GENO_BLOCK - 1) / GENO_BLOCK, n), dim3(GENO_BLOCK), 0, stream, G, d_tables);
		++launches;
		GENO_DEV(hipGetLastError());
	}
	// ---- pass 1: B_{c-1} for c = n-1 .. 1, kept where c - 1 is the last column of a window (nothing to keep with one window)
	if (n_windows > 1) {
		const Buf* in = nullptr;
		uint32_t flip = 0;
		for (uint32_t c = n - 1; c >= 1; --c) {
			const bool keep = (c - 1) % K == K - 1;
			Buf& out = keep ? ckpt[(c - 1) / K] : pp[flip];
			GENO_DEV(backward(c, in, out));
			in = &out;
			if (!keep) flip ^= 1u;
		}
	}
	GENO_DEV(hipEventRecord(ev[1], stream));
	auto fwd_args = [&](uint32_t c, const Buf* prev_alpha, const Buf* beta, Buf* out, double* glp) {
		const bool last = c + 1 == n;
		const uint32_t fc = last ? 0u : p.f[c], fm = last ? 0u : p.fwd_mask[c];
		const uint32_t atomics = (!last && out && (uint32_t)p.k[c] - fc > GENO_LOOP_BITS) ? 1u : 0u;
		GenoFwdArgs a{};
		a.C = GenoCol{c, p.k[c], p.b[c], fc, fm, std::min<uint32_t>((uint32_t)p.k[c] - fc, GENO_LOOP_BITS), atomics, 0u};
		a.prev = prev_alpha ? prev_alpha->v : nullptr; a.prev_partials = prev_alpha ? prev_alpha->partials : nullptr;
		a.prev_blocks = prev_alpha ? prev_alpha->blocks : 0u;
		a.beta = beta ? beta->v : nullptr; a.beta_partials = beta ? beta->partials : nullptr; a.beta_blocks = beta ? beta->blocks : 0u;
		a.out = (last || !out) ? nullptr : out->v; a.out_partials = out ? out->partials : nullptr;
		a.gl_partials = glp;
		a.n_blocks = fw_blocks[c];
		return a;
	};
	auto launch_forward = [&](const GenoFwdArgs& a, int mode, hipStream_t on) -> hipError_t {
		if (a.C.use_atomics) { hipError_t e = hipMemsetAsync(a.out, 0, ((size_t)T << a.C.f) * 8, on); if (e != hipSuccess) return e; }
		const dim3 grid(a.n_blocks), block(GENO_BLOCK);
#define GENO_FWD(TT) do { if (mode == 0) hipLaunchKernelGGL((geno_forward<TT, 0>), grid, block, table_bytes, on, G, a, (const GenoFwdArgs*)nullptr); \
		                     else hipLaunchKernelGGL((geno_forward<TT, 1>), grid, block, table_bytes, on, G, a, (const GenoFwdArgs*)nullptr); } while (0)
		if (T == 1) GENO_FWD(1); else if (T == 4) GENO_FWD(4); else GENO_FWD(16);
#undef GENO_FWD
		++launches;
		return hipGetLastError();
	};
	if (n_windows == 1) {
		// ---- everything fits: the backward chain (this stream) and the forward chain of the A columns (a second stream) run side
		// by side -- they meet only in the likelihood sums, which one batched launch per 1024 columns computes afterwards
		hipStream_t stream2 = nullptr;
		GENO_DEV(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
		extra_streams.push_back(stream2);
		hipEvent_t ev_start, ev_fwd;
		GENO_DEV(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
		all_events.push_back(ev_start);
		GENO_DEV(hipEventCreateWithFlags(&ev_fwd, hipEventDisableTiming));
		all_events.push_back(ev_fwd);
		GENO_DEV(hipEventRecord(ev_start, stream));
		GENO_DEV(hipStreamWaitEvent(stream2, ev_start, 0));   // (uploads happened on `stream`)
		// interleave the submissions so that neither hardware queue runs dry
		uint32_t cb = n - 1, cf = 0;
		while (cb >= 1 || cf + 1 < n) {
			if (cb >= 1) {
				GENO_DEV(backward(cb, cb == n - 1 ? nullptr : &wstore[cb], wstore[cb - 1]));
				--cb;
			}
			if (cf + 1 < n) {
				astore[cf].blocks = fw_blocks[cf];
				GENO_DEV(launch_forward(fwd_args(cf, cf ? &astore[cf - 1] : nullptr, nullptr, &astore[cf], nullptr), 1, stream2));
				++cf;
			}
		}
		GENO_DEV(hipEventRecord(ev_fwd, stream2));
		GENO_DEV(hipStreamWaitEvent(stream, ev_fwd, 0));
		GENO_DEV(hipEventRecord(ev[1], stream));
		constexpr uint32_t BATCH = 1024;
		std::vector<GenoFwdArgs> batch(n);
		for (uint32_t c = 0; c < n; ++c)
			batch[c] = fwd_args(c, c ? &astore[c - 1] : nullptr, c + 1 < n ? &wstore[c] : nullptr, nullptr, d_glpart + (size_t)(c % BATCH) * max_blocks * n_gl);
		void* d_batch = nullptr;
		GENO_DEV(up(&d_batch, batch.data(), batch.size() * sizeof(GenoFwdArgs)));
		for (uint32_t c0 = 0; c0 < n; c0 += BATCH) {
			const uint32_t cols = std::min(BATCH, n - c0);
			uint32_t gx = 1;
			for (uint32_t c = c0; c < c0 + cols; ++c) gx = std::max(gx, fw_blocks[c]);
			const dim3 grid(gx, cols), block(GENO_BLOCK);
			const GenoFwdArgs none{};
			const GenoFwdArgs* bp = (const GenoFwdArgs*)d_batch + c0;
			if (T == 1) hipLaunchKernelGGL((geno_forward<1, 2>), grid, block, table_bytes, stream, G, none, bp);
			else if (T == 4) hipLaunchKernelGGL((geno_forward<4, 2>), grid, block, table_bytes, stream, G, none, bp);
			else hipLaunchKernelGGL((geno_forward<16, 2>), grid, block, table_bytes, stream, G, none, bp);
			++launches;
			GENO_DEV(hipGetLastError());
			hipLaunchKernelGGL(geno_finish, dim3(cols), dim3(64), 0, stream, d_glpart, (const uint32_t*)d_fwb, c0, max_blocks, ni, n, d_gl);
			++launches;
			GENO_DEV(hipGetLastError());
		}
		GENO_DEV(hipStreamSynchronize(stream));
	} else {
	// ---- windows: the backward columns of window w + 1 are recomputed on a second stream (into the other half of the window
	// store) while the forward pass runs through window w
	hipStream_t stream2 = nullptr;
	GENO_DEV(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
	extra_streams.push_back(stream2);
	hipEvent_t ev_back[2], ev_fwd[2], ev_pass1;
	for (hipEvent_t& e : ev_back) { GENO_DEV(hipEventCreateWithFlags(&e, hipEventDisableTiming)); all_events.push_back(e); }
	for (hipEvent_t& e : ev_fwd) { GENO_DEV(hipEventCreateWithFlags(&e, hipEventDisableTiming)); all_events.push_back(e); }
	GENO_DEV(hipEventCreateWithFlags(&ev_pass1, hipEventDisableTiming));
	all_events.push_back(ev_pass1);
	GENO_DEV(hipEventRecord(ev_pass1, stream));
	GENO_DEV(hipStreamWaitEvent(stream2, ev_pass1, 0));   // the kept columns (and the uploads) are complete
	auto recompute = [&](uint32_t w) -> hipError_t {      // B_c for the columns of window w but its last, on stream2
		const uint32_t lo = w * K, hi = std::min(n, lo + K);
		Buf* half = wstore.data() + (size_t)(w & 1u) * K;
		const Buf* last_beta = hi == n ? nullptr : &ckpt[w];   // B_{hi-1}
		for (uint32_t c = hi - 1; c > lo; --c) {
			const Buf* in = c == hi - 1 ? last_beta : &half[c - lo];
			hipError_t e = backward(c, in, half[c - 1 - lo], stream2);
			if (e != hipSuccess) return e;
		}
		return hipEventRecord(ev_back[w & 1u], stream2);
	};
	GENO_DEV(recompute(0));
	uint32_t aflip = 0;
	const Buf* prev_alpha = nullptr;
	for (uint32_t w = 0; w < n_windows; ++w) {
		const uint32_t lo = w * K, hi = std::min(n, lo + K);
		const Buf* half = wstore.data() + (size_t)(w & 1u) * K;
		const Buf* last_beta = hi == n ? nullptr : &ckpt[w];
		GENO_DEV(hipStreamWaitEvent(stream, ev_back[w & 1u], 0));
		if (w + 1 < n_windows) {
			if (w >= 1) GENO_DEV(hipStreamWaitEvent(stream2, ev_fwd[(w - 1) & 1u], 0));   // the forward pass of window w - 1 is done with that half
			GENO_DEV(recompute(w + 1));
		}
		for (uint32_t c = lo; c < hi; ++c) {
			const Buf* beta = c == hi - 1 ? last_beta : &half[c - lo];
			Buf& out = alpha[aflip];
			out.blocks = fw_blocks[c];
			GENO_DEV(launch_forward(fwd_args(c, prev_alpha, beta, &out, d_glpart + (size_t)(c - lo) * max_blocks * n_gl), 0, stream));
			prev_alpha = &out;
			aflip ^= 1u;
		}
		hipLaunchKernelGGL(geno_finish, dim3(hi - lo), dim3(64), 0, stream, d_glpart, (const uint32_t*)d_fwb, lo, max_blocks, ni, n, d_gl);
		++launches;
		GENO_DEV(hipGetLastError());
		GENO_DEV(hipEventRecord(ev_fwd[w & 1u], stream));
	}
	GENO_DEV(hipStreamSynchronize(stream));
	GENO_DEV(hipStreamSynchronize(stream2));
	}
	GENO_DEV(hipEventRecord(ev[2], stream));
	if (getenv("WHAMD_DEBUG_TIMING")) {
		const double enq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enqueue0).count();
		GENO_DEV(hipStreamSynchronize(stream));
		const double all = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enqueue0).count();
		fprintf(stderr, "[whamd timing] genotype: %llu launches submitted in %.1f ms (host), stream drained after %.1f ms\n", (unsigned long long)launches, enq, all);
	}
	GENO_DEV(hipMemcpyAsync(gl_out.data(), d_gl, gl_out.size() * 8, hipMemcpyDeviceToHost, stream));
	GENO_DEV(hipStreamSynchronize(stream));
	float ms01 = 0, ms12 = 0, ms02 = 0;
	GENO_DEV(hipEventElapsedTime(&ms01, ev[0], ev[1]));
	GENO_DEV(hipEventElapsedTime(&ms12, ev[1], ev[2]));
	GENO_DEV(hipEventElapsedTime(&ms02, ev[0], ev[2]));
	st.backward_ms = ms01;
	st.forward_ms = ms12;
	st.total_ms = ms02;
	st.launches = launches;
	const double ms_done = phase_ms();
	cleanup();
	if (getenv("WHAMD_DEBUG_TIMING"))
		fprintf(stderr, "[whamd timing] genotype phases (wall): allocations + uploads %.1f ms, submission + device %.1f ms, freeing %.1f ms\n",
		        ms_allocated, ms_done - ms_allocated, phase_ms() - ms_done);
#undef GENO_DEV
	return WHAMD_OK;
}

}  // namespace whamd
