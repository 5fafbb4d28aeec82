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
#include <cmath>
#include <cstdio>
#include <cstring>

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
};

__device__ __forceinline__ void geno_stage(const GenoDev& G, uint32_t c, uint32_t k, GenoShared& S) {
	const uint32_t tid = threadIdx.x;
	const uint64_t e0 = G.col_ptr[c];
	if (tid < k) { S.pe[tid] = G.ent_pe[e0 + tid]; S.ind[tid] = G.ent_ind[e0 + tid]; S.allele[tid] = G.ent_allele[e0 + tid]; }
	for (uint32_t i = tid; i < G.T * G.A; i += GENO_BLOCK) S.prior[i] = G.prior[(size_t)c * G.T * G.A + i];
	for (uint32_t i = tid; i < G.T * G.A * G.n_ind; i += GENO_BLOCK) S.gidx[i] = G.gidx[i];
	for (uint32_t i = tid; i < G.T * G.n_ind * 2; i += GENO_BLOCK) S.h2p[i] = G.h2p[i];
	if (tid < G.nb) S.bern[tid] = G.bern[(size_t)c * G.nb + tid];
}

// 1 / (sum of the per-block sums of a stored column), the same in every thread of every reader
__device__ __forceinline__ double geno_inverse_total(const double* partials, uint32_t n_blocks, double* red) {
	double v = 0.0;
	for (uint32_t i = threadIdx.x; i < n_blocks; i += GENO_BLOCK) v += partials[i];
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
	if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	double total = 0.0;
	for (int w = 0; w < GENO_BLOCK / 64; ++w) total += red[w];
	__syncthreads();
	return total > 0.0 ? 1.0 / total : 0.0;
}

// W_i(x) of one cell (see the header of this file): the reads of the column in LDS
__device__ __forceinline__ void geno_partition_products(const GenoDev& G, const GenoShared& S, uint32_t k, uint32_t x, uint32_t i, double (&W)[4][2]) {
#pragma unroll
	for (int p = 0; p < 4; ++p) W[p][0] = W[p][1] = 1.0;
	for (uint32_t j = 0; j < k; ++j) {
		const uint32_t al = S.allele[j];
		if (al > 1u) continue;                       // BLANK
		const uint32_t bit = (x >> j) & 1u;
		// bit 0 <-> "entry_in_partition1" (src/genotypecolumncostcomputer.cpp:61): haplotype 1 of the read's individual
		const uint32_t part = (uint32_t)S.h2p[((size_t)i * G.n_ind + S.ind[j]) * 2 + (bit ^ 1u)];
		const double pe = S.pe[j], ok = 1.0 - pe;
		const double m0 = al == 0u ? ok : pe, m1 = al == 0u ? pe : ok;   // factor for "the partition carries allele 0 / 1"
#pragma unroll
		for (int p = 0; p < 4; ++p) {
			W[p][0] *= part == (uint32_t)p ? m0 : 1.0;
			W[p][1] *= part == (uint32_t)p ? m1 : 1.0;
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

// Backward step of column c: reads B_c (`in`, null for the last column), writes B_{c-1} (`out`, 2^b_c x T) and the per-block sums.
template <int T>
__global__ __launch_bounds__(GENO_BLOCK) void geno_backward(GenoDev G, uint32_t c, const double* __restrict__ in, const double* __restrict__ in_partials,
                                                           uint32_t in_blocks, double* __restrict__ out, double* __restrict__ out_partials, uint32_t use_atomics) {
	__shared__ GenoShared S;
	const uint32_t k = G.k[c], b = G.b[c], fmask = G.fwd_mask[c];
	geno_stage(G, c, k, S);
	__syncthreads();
	const double inv = in ? geno_inverse_total(in_partials, in_blocks, &S.red[0][0]) : 1.0;
	const uint32_t nfree = k - b, loop_bits = nfree < GENO_LOOP_BITS ? nfree : GENO_LOOP_BITS;
	const uint64_t n_threads = 1ull << (k - loop_bits);
	const uint64_t t = (uint64_t)blockIdx.x * GENO_BLOCK + threadIdx.x;
	double acc[T];
#pragma unroll
	for (int j = 0; j < T; ++j) acc[j] = 0.0;
	uint32_t y = 0;
	if (t < n_threads) {
		y = (uint32_t)t & ((1u << b) - 1u);
		const uint32_t chunk = (uint32_t)(t >> b);
		for (uint32_t e = 0; e < (1u << loop_bits); ++e) {
			const uint32_t x = y | (((chunk << loop_bits) | e) << b);
			const uint32_t yf = in ? geno_pext(x, fmask) : 0u;
#pragma unroll
			for (int i = 0; i < T; ++i) {
				const double beta = in ? in[(size_t)yf * T + i] * inv : 1.0;
				double W[4][2];
				geno_partition_products(G, S, k, x, (uint32_t)i, W);
				double s = 0.0;
				for (uint32_t a = 0; a < G.A; ++a) s += S.prior[i * G.A + a] * geno_assignment_cost(W, G.P, a);
				s *= beta;
#pragma unroll
				for (int j = 0; j < T; ++j) acc[j] += s * S.bern[__popc((uint32_t)(i ^ j))];
			}
		}
		if (use_atomics) {
#pragma unroll
			for (int j = 0; j < T; ++j) atomicAdd(out + (size_t)y * T + j, acc[j]);
		} else {
#pragma unroll
			for (int j = 0; j < T; ++j) out[(size_t)y * T + j] = acc[j];
		}
	}
	// per-block sum of what was written
	double v = 0.0;
#pragma unroll
	for (int j = 0; j < T; ++j) v += acc[j];
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
template <int T>
__global__ __launch_bounds__(GENO_BLOCK) void geno_forward(GenoDev G, uint32_t c, const double* __restrict__ prev, const double* __restrict__ prev_partials,
                                                          uint32_t prev_blocks, const double* __restrict__ beta, const double* __restrict__ beta_partials,
                                                          uint32_t beta_blocks, double* __restrict__ out, double* __restrict__ out_partials,
                                                          double* __restrict__ gl_partials, uint32_t use_atomics) {
	__shared__ GenoShared S;
	const uint32_t k = G.k[c], b = G.b[c], f = out ? G.f[c] : 0u, fmask = out ? G.fwd_mask[c] : 0u;
	geno_stage(G, c, k, S);
	__syncthreads();
	const double inv_prev = prev ? geno_inverse_total(prev_partials, prev_blocks, &S.red[0][0]) : 1.0;
	const double inv_beta = beta ? geno_inverse_total(beta_partials, beta_blocks, &S.red[0][0]) : 1.0;
	const uint32_t kmask = k >= 32u ? 0xFFFFFFFFu : ((1u << k) - 1u), endmask = kmask & ~fmask;
	const uint32_t nfree = k - f, loop_bits = nfree < GENO_LOOP_BITS ? nfree : GENO_LOOP_BITS;
	const uint64_t n_threads = 1ull << (k - loop_bits);
	const uint64_t t = (uint64_t)blockIdx.x * GENO_BLOCK + threadIdx.x;
	const uint32_t n_gl = 1u + 3u * G.n_ind;
	double gl[GENO_MAXGL];
#pragma unroll
	for (int q = 0; q < GENO_MAXGL; ++q) gl[q] = 0.0;
	double acc[T];
#pragma unroll
	for (int i = 0; i < T; ++i) acc[i] = 0.0;
	if (t < n_threads) {
		const uint32_t yf = (uint32_t)t & ((1u << f) - 1u), chunk = (uint32_t)(t >> f);
		const uint32_t xf = geno_pdep(yf, fmask);
		// the genotype likelihood the true B_c of the LAST window column may be absent (last column of the table): beta = 1
		for (uint32_t e = 0; e < (1u << loop_bits); ++e) {
			const uint32_t x = xf | geno_pdep((chunk << loop_bits) | e, endmask);
			const uint32_t yb = x & ((1u << b) - 1u);
#pragma unroll
			for (int i = 0; i < T; ++i) {
				double sum_prev = 1.0;
				if (prev) {
					sum_prev = 0.0;
#pragma unroll
					for (int j = 0; j < T; ++j) sum_prev += prev[(size_t)yb * T + j] * S.bern[__popc((uint32_t)(i ^ j))];
					sum_prev *= inv_prev;
				}
				const double bt = beta ? beta[(size_t)yf * T + i] * inv_beta : 1.0;
				double W[4][2];
				geno_partition_products(G, S, k, x, (uint32_t)i, W);
				for (uint32_t a = 0; a < G.A; ++a) {
					const double fw = sum_prev * geno_assignment_cost(W, G.P, a) * S.prior[i * G.A + a];
					const double fb = fw * bt;
					acc[i] += fw;
					gl[0] += fb;
					const uint8_t* gi = S.gidx + ((size_t)i * G.A + a) * G.n_ind;
#pragma unroll
					for (int s = 0; s < MAX_IND; ++s) {
						if ((uint32_t)s < G.n_ind) {
							const uint32_t g = gi[s];
							gl[1 + 3 * s + 0] += g == 0u ? fb : 0.0;
							gl[1 + 3 * s + 1] += g == 1u ? fb : 0.0;
							gl[1 + 3 * s + 2] += g == 2u ? fb : 0.0;
						}
					}
				}
			}
		}
		if (out) {
			if (use_atomics) {
#pragma unroll
				for (int i = 0; i < T; ++i) atomicAdd(out + (size_t)yf * T + i, acc[i]);
			} else {
#pragma unroll
				for (int i = 0; i < T; ++i) out[(size_t)yf * T + i] = acc[i];
			}
		}
	}
	// per-block sums: the written column (slot n_gl) and the likelihood numerators
	double total_out = 0.0;
#pragma unroll
	for (int i = 0; i < T; ++i) total_out += acc[i];
	__syncthreads();
#pragma unroll
	for (int q = 0; q <= GENO_MAXGL; ++q) {
		if ((uint32_t)q <= n_gl) {   // (wave-uniform)
			double v = (uint32_t)q == n_gl ? total_out : gl[q < GENO_MAXGL ? q : 0];
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

uint32_t blocks_for(uint32_t k, uint32_t proj) {   // grid of a column kernel: 2^(k - min(k - proj, LOOP)) threads
	const uint32_t nfree = k - proj, loop_bits = std::min(nfree, GENO_LOOP_BITS);
	const uint64_t threads = 1ull << (k - loop_bits);
	return (uint32_t)((threads + GENO_BLOCK - 1) / GENO_BLOCK);
}

}  // namespace

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
	uint32_t max_k = 0, max_proj = 0;
	for (uint32_t c = 0; c < n; ++c) {
		max_k = std::max<uint32_t>(max_k, p.k[c]);
		max_proj = std::max<uint32_t>(max_proj, std::max<uint32_t>(p.f[c], p.b[c]));
		st.n_cells += 1ull << p.k[c];
	}
	st.max_coverage = max_k;
	uint32_t K = window_hint ? window_hint : (uint32_t)std::ceil(std::sqrt((double)n));
	K = std::max(1u, std::min(K, n));
	st.window = K;
	const uint32_t n_windows = (n + K - 1) / K;
	const size_t buf_doubles = ((size_t)1 << max_proj) * T;
	const uint32_t max_blocks = (uint32_t)((((size_t)1 << max_k) + GENO_BLOCK - 1) / GENO_BLOCK);
	const uint32_t n_gl = 1 + 3 * ni;
	{
		size_t free_b = 0, total_b = 0;
		GENO_TRY(hipMemGetInfo(&free_b, &total_b));
		const double need = (double)(buf_doubles * 8 + (size_t)max_blocks * 8) * (n_windows + K + 4.0) + (double)K * max_blocks * n_gl * 8 + (double)p.entries.size() * 10 + (double)n * (64 + 8.0 * T * m.A);
		if (need + (double)(1ull << 30) > (double)free_b) {
			msg = "genotyping buffers of " + std::to_string((uint64_t)(need / 1048576.0)) + " MiB do not fit in free HBM";
			return WHAMD_ERR_UNSUPPORTED;
		}
	}
	hipStream_t stream = nullptr;
	GENO_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
	std::vector<void*> allocations;
	auto cleanup = [&]() {
		for (void* a : allocations) (void)hipFree(a);
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
		fw_blocks[c] = blocks_for(p.k[c], c + 1 < n ? p.f[c] : 0u);
		bw_blocks[c] = blocks_for(p.k[c], p.b[c]);
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
	// ---- buffers: every column buffer carries its per-block sums
	struct Buf { double* v = nullptr; double* partials = nullptr; uint32_t blocks = 0; };
	auto make_buf = [&](Buf& bf) -> hipError_t {
		hipError_t e = alloc((void**)&bf.v, buf_doubles * 8);
		if (e == hipSuccess) e = alloc((void**)&bf.partials, (size_t)max_blocks * 8);
		return e;
	};
	Buf alpha[2], pp[2];
	std::vector<Buf> ckpt(n_windows), wstore(K);
	for (Buf& bf : alpha) GENO_DEV(make_buf(bf));
	for (Buf& bf : pp) GENO_DEV(make_buf(bf));
	for (Buf& bf : ckpt) GENO_DEV(make_buf(bf));
	for (Buf& bf : wstore) GENO_DEV(make_buf(bf));
	double *d_glpart = nullptr, *d_gl = nullptr;
	GENO_DEV(alloc((void**)&d_glpart, (size_t)K * max_blocks * n_gl * 8));
	GENO_DEV(alloc((void**)&d_gl, gl_out.size() * 8));
	hipEvent_t ev[3];
	for (hipEvent_t& e : ev) GENO_DEV(hipEventCreate(&e));
	uint64_t launches = 0;
	// one backward step: column c, B_c in `in` (null: last column) -> B_{c-1} in `out`
	auto backward = [&](uint32_t c, const Buf* in, Buf& out) -> hipError_t {
		const uint32_t blocks = bw_blocks[c];
		const uint32_t atomics = (uint32_t)p.k[c] - p.b[c] > GENO_LOOP_BITS ? 1u : 0u;
		if (atomics) { hipError_t e = hipMemsetAsync(out.v, 0, ((size_t)T << p.b[c]) * 8, stream); if (e != hipSuccess) return e; }
		out.blocks = blocks;
		const double* iv = in ? in->v : nullptr;
		const double* ip = in ? in->partials : nullptr;
		const uint32_t ib = in ? in->blocks : 0u;
		if (T == 1) hipLaunchKernelGGL(geno_backward<1>, dim3(blocks), dim3(GENO_BLOCK), 0, stream, G, c, iv, ip, ib, out.v, out.partials, atomics);
		else if (T == 4) hipLaunchKernelGGL(geno_backward<4>, dim3(blocks), dim3(GENO_BLOCK), 0, stream, G, c, iv, ip, ib, out.v, out.partials, atomics);
		else hipLaunchKernelGGL(geno_backward<16>, dim3(blocks), dim3(GENO_BLOCK), 0, stream, G, c, iv, ip, ib, out.v, out.partials, atomics);
		++launches;
		return hipGetLastError();
	};
	GENO_DEV(hipEventRecord(ev[0], stream));
	// ---- pass 1: B_{c-1} for c = n-1 .. 1, kept where c - 1 is the last column of a window
	{
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
	// ---- windows: recompute the backward columns of the window, then the forward pass through it
	uint32_t aflip = 0;
	const Buf* prev_alpha = nullptr;
	for (uint32_t w = 0; w < n_windows; ++w) {
		const uint32_t lo = w * K, hi = std::min(n, lo + K);
		const Buf* last_beta = hi == n ? nullptr : &ckpt[w];   // B_{hi-1}
		for (uint32_t c = hi - 1; c > lo; --c) {
			const Buf* in = c == hi - 1 ? last_beta : &wstore[c - lo];
			GENO_DEV(backward(c, in, wstore[c - 1 - lo]));
		}
		for (uint32_t c = lo; c < hi; ++c) {
			const Buf* beta = c == hi - 1 ? last_beta : &wstore[c - lo];
			const bool last = c + 1 == n;
			Buf& out = alpha[aflip];
			const uint32_t blocks = fw_blocks[c];
			const uint32_t atomics = (!last && (uint32_t)p.k[c] - p.f[c] > GENO_LOOP_BITS) ? 1u : 0u;
			if (atomics) GENO_DEV(hipMemsetAsync(out.v, 0, ((size_t)T << p.f[c]) * 8, stream));
			out.blocks = blocks;
			double* glp = d_glpart + (size_t)(c - lo) * max_blocks * n_gl;
			const double *pv = prev_alpha ? prev_alpha->v : nullptr, *ppart = prev_alpha ? prev_alpha->partials : nullptr;
			const uint32_t pb = prev_alpha ? prev_alpha->blocks : 0u;
			const double *bv = beta ? beta->v : nullptr, *bpart = beta ? beta->partials : nullptr;
			const uint32_t bb = beta ? beta->blocks : 0u;
			double* ov = last ? nullptr : out.v;
			if (T == 1) hipLaunchKernelGGL(geno_forward<1>, dim3(blocks), dim3(GENO_BLOCK), 0, stream, G, c, pv, ppart, pb, bv, bpart, bb, ov, out.partials, glp, atomics);
			else if (T == 4) hipLaunchKernelGGL(geno_forward<4>, dim3(blocks), dim3(GENO_BLOCK), 0, stream, G, c, pv, ppart, pb, bv, bpart, bb, ov, out.partials, glp, atomics);
			else hipLaunchKernelGGL(geno_forward<16>, dim3(blocks), dim3(GENO_BLOCK), 0, stream, G, c, pv, ppart, pb, bv, bpart, bb, ov, out.partials, glp, atomics);
			++launches;
			GENO_DEV(hipGetLastError());
			prev_alpha = &out;
			aflip ^= 1u;
		}
		hipLaunchKernelGGL(geno_finish, dim3(hi - lo), dim3(64), 0, stream, d_glpart, (const uint32_t*)d_fwb, lo, max_blocks, ni, n, d_gl);
		++launches;
		GENO_DEV(hipGetLastError());
	}
	GENO_DEV(hipEventRecord(ev[2], stream));
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
	for (hipEvent_t& e : ev) (void)hipEventDestroy(e);
	cleanup();
#undef GENO_DEV
	return WHAMD_OK;
}

}  // namespace whamd
