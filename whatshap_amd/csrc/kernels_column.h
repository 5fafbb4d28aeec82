// kernels_column.h -- per-column kernels (one launch per column): column_step_fused, column_step_keys, column_finalize, and the device helpers every kernel shares (bit deposit, Gray rank, LDS staging of a column).
// Included by dp_device.hip inside namespace whamd { namespace { ... } }: not a stand-alone header.

constexpr int QMAX = 4;  // a thread enumerates at most 2^QMAX ending-bit patterns itself

// Bit deposit through host-precomputed runs of the mask (gfx950 has no PDEP): the segment list is wave-uniform.
__device__ __forceinline__ uint32_t deposit(uint32_t v, const uint32_t* __restrict__ segs, uint32_t nseg) {
	uint32_t x = 0;
	for (uint32_t i = 0; i < nseg; ++i) {
		const uint32_t sg = segs[i];
		const uint32_t src = sg & 31u, dst = (sg >> 8) & 31u, len = (sg >> 16) & 31u;
		x |= ((v >> src) & ((1u << len) - 1u)) << dst;
	}
	return x;
}

// Position of x in the reflected Gray code sequence g(r) = r ^ (r >> 1)  (src/graycodes.cpp:26-43 visits g(0), g(1), ...).
__device__ __forceinline__ uint32_t gray_rank(uint32_t x) {
	x ^= x >> 1;
	x ^= x >> 2;
	x ^= x >> 4;
	x ^= x >> 8;
	x ^= x >> 16;
	return x;
}

// Per-launch staging of a column's cost data in LDS: 5-bit lookup tables of the per-individual sums L_s(x) and the
// term lists of every transmission value, so that a cell costs ceil(k/5) LDS reads per individual instead of a
// k-step loop over scalar global loads.
constexpr int COL_CHUNKS = 5, COL_MAXTERMS = 1024;
template <int T, int NIND>
struct ColumnStage {
	int32_t lut[NIND][COL_CHUNKS][32];
	DevTerm terms[COL_MAXTERMS];
	uint32_t tptr[T + 1];
};

template <int T, int NIND>
__device__ __forceinline__ void stage_column(ColumnStage<T, NIND>& S, const DevProblem& P, const DevColumn& col) {
	const int32_t* __restrict__ dl = P.delta + col.delta_off;
	const uint32_t k = col.k;
	for (uint32_t idx = threadIdx.x; idx < (uint32_t)(NIND * COL_CHUNKS * 32); idx += blockDim.x) {
		const uint32_t s = idx / (COL_CHUNKS * 32), chunk = (idx / 32) % COL_CHUNKS, v = idx & 31u;
		int32_t sum = 0;
#pragma unroll
		for (int j = 0; j < 5; ++j) {
			const uint32_t bit = chunk * 5 + j;
			if (bit < k && ((v >> j) & 1u)) sum += dl[s * k + bit];
		}
		S.lut[s][chunk][v] = sum;
	}
	const uint32_t* __restrict__ tp = P.term_ptr + col.term_off;
	const uint32_t t0 = tp[0], nterms = min(tp[T] - t0, (uint32_t)COL_MAXTERMS);
	for (uint32_t i = threadIdx.x; i < nterms; i += blockDim.x) S.terms[i] = P.terms[t0 + i];
	if (threadIdx.x <= (uint32_t)T) S.tptr[threadIdx.x] = min(tp[threadIdx.x] - t0, (uint32_t)COL_MAXTERMS);
	__syncthreads();
}

// cost_{c,t}(x) for all t (get_cost, src/pedigreecolumncostcomputer.cpp:101-114) from the per-individual sums L_s(x).
template <int T, int NIND>
__device__ __forceinline__ void cell_costs(uint32_t x, const ColumnStage<T, NIND>& S, uint32_t nchunks, uint32_t (&cost)[T]) {
	int32_t L[NIND];
#pragma unroll
	for (int s = 0; s < NIND; ++s) L[s] = 0;
	for (uint32_t c = 0; c < nchunks; ++c) {
		const uint32_t v = (x >> (5 * c)) & 31u;
#pragma unroll
		for (int s = 0; s < NIND; ++s) L[s] += S.lut[s][c][v];
	}
#pragma unroll
	for (int t = 0; t < T; ++t) {
		uint32_t best = 0xFFFFFFFFu;
		const uint32_t e = S.tptr[t + 1];
		for (uint32_t q = S.tptr[t]; q < e; ++q) {
			const DevTerm tm = S.terms[q];
			uint32_t v = tm.c;
#pragma unroll
			for (int s = 0; s < NIND; ++s) {
				v += ((tm.plus >> s) & 1u) ? (uint32_t)L[s] : 0u;
				v -= ((tm.minus >> s) & 1u) ? (uint32_t)L[s] : 0u;
			}
			best = min(best, v);
		}
		cost[t] = best;
	}
}

// D[i] and argj[i] of one cell (src/pedigreedptable.cpp:264-300).  prev == nullptr for column 0.
template <int T>
__device__ __forceinline__ void cell_dp(const uint32_t (&cost)[T], const uint32_t* __restrict__ prev, uint32_t z,
                                        uint32_t recomb, uint32_t (&D)[T], uint32_t (&aj)[T]) {
	uint32_t pv[T];
	if (prev) {
		if constexpr (T == 1) {
			pv[0] = prev[z];
		} else {
			const uint4* p4 = reinterpret_cast<const uint4*>(prev + (size_t)z * T);
#pragma unroll
			for (int q = 0; q < T / 4; ++q) {
				const uint4 v = p4[q];
				pv[4 * q] = v.x; pv[4 * q + 1] = v.y; pv[4 * q + 2] = v.z; pv[4 * q + 3] = v.w;
			}
		}
	} else {
#pragma unroll
		for (int j = 0; j < T; ++j) pv[j] = 0;
	}
#pragma unroll
	for (int i = 0; i < T; ++i) {
		uint32_t m = 0xFFFFFFFFu, mj = 0;
		if (cost[i] != 0xFFFFFFFFu) {
#pragma unroll
			for (int j = 0; j < T; ++j) {
				if (pv[j] != 0xFFFFFFFFu) {
					const uint32_t val = cost[i] + pv[j] + (uint32_t)__popc((unsigned)(i ^ j)) * recomb;
					if (val < m) { m = val; mj = j; }
				}
			}
		}
		D[i] = m;
		aj[i] = mj;
	}
}

template <int T, int NIND>
__global__ __launch_bounds__(256) void column_step_fused(DevProblem P, uint32_t c, const uint32_t* __restrict__ prev,
                                                          uint32_t* __restrict__ cur) {
	const DevColumn col = P.cols[c];
	__shared__ ColumnStage<T, NIND> stage;
	stage_column<T, NIND>(stage, P, col);
	const uint32_t nchunks = (col.k + 4) / 5;
	const uint32_t y = blockIdx.x * blockDim.x + threadIdx.x;  // grid covers exactly 2^f entries (f >= 6)
	const uint32_t* __restrict__ segs = P.segs + col.seg_off;
	const uint32_t xbase = deposit(y, segs, col.nseg_fwd);
	const uint32_t lowmask = (1u << col.b) - 1u;
	const uint32_t* pr = c ? prev : nullptr;
	uint32_t bD[T], bR[T], bV[T];
#pragma unroll
	for (int i = 0; i < T; ++i) { bD[i] = 0xFFFFFFFFu; bR[i] = 0xFFFFFFFFu; bV[i] = 0; }
	const uint32_t ne = 1u << col.ebits;
	for (uint32_t e = 0; e < ne; ++e) {
		const uint32_t x = xbase | deposit(e, segs + col.nseg_fwd, col.nseg_end);
		uint32_t cost[T], D[T], aj[T];
		cell_costs<T, NIND>(x, stage, nchunks, cost);
		cell_dp<T>(cost, pr, x & lowmask, col.recomb, D, aj);
		const uint32_t r = gray_rank(x);
#pragma unroll
		for (int i = 0; i < T; ++i) {
			const bool better = (D[i] < bD[i]) || (D[i] == bD[i] && r < bR[i]);
			if (better) { bD[i] = D[i]; bR[i] = r; bV[i] = e | (aj[i] << col.ebits); }
		}
	}
	if constexpr (T == 1) {
		cur[y] = bD[0];
	} else {
		uint4* c4 = reinterpret_cast<uint4*>(cur + (size_t)y * T);
#pragma unroll
		for (int q = 0; q < T / 4; ++q) c4[q] = make_uint4(bD[4 * q], bD[4 * q + 1], bD[4 * q + 2], bD[4 * q + 3]);
	}
	// backtrace record: nplanes bit planes per transmission value, one ballot word per 64 consecutive y
	unsigned long long* planes = reinterpret_cast<unsigned long long*>(P.bt + col.bt_off);
	const uint32_t words = 1u << (col.f - 6);
	const uint32_t w = y >> 6;
	for (uint32_t p = 0; p < col.nplanes; ++p) {
#pragma unroll
		for (int i = 0; i < T; ++i) {
			const unsigned long long word = __ballot((bV[i] >> p) & 1u);
			if ((threadIdx.x & 63u) == 0) planes[(size_t)(p * T + i) * words + w] = word;
		}
	}
}

template <int T, int NIND>
__global__ __launch_bounds__(256) void column_step_keys(DevProblem P, uint32_t c, const uint32_t* __restrict__ prev,
                                                         uint32_t total_threads) {
	const DevColumn col = P.cols[c];
	__shared__ ColumnStage<T, NIND> stage;
	stage_column<T, NIND>(stage, P, col);
	const uint32_t nchunks = (col.k + 4) / 5;
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= total_threads) return;
	const uint32_t y = gid & ((1u << col.f) - 1u);
	const uint32_t chunk = gid >> col.f;
	const uint32_t* __restrict__ segs = P.segs + col.seg_off;
	const uint32_t xbase = deposit(y, segs, col.nseg_fwd);
	const uint32_t lowmask = (1u << col.b) - 1u;
	const uint32_t* pr = c ? prev : nullptr;
	unsigned long long best[T];
#pragma unroll
	for (int i = 0; i < T; ++i) best[i] = ~0ull;
	const uint32_t ne = 1u << col.eloop;
	for (uint32_t el = 0; el < ne; ++el) {
		const uint32_t e = (chunk << col.eloop) | el;
		const uint32_t x = xbase | deposit(e, segs + col.nseg_fwd, col.nseg_end);
		uint32_t cost[T], D[T], aj[T];
		cell_costs<T, NIND>(x, stage, nchunks, cost);
		cell_dp<T>(cost, pr, x & lowmask, col.recomb, D, aj);
		const uint32_t r = gray_rank(x);
#pragma unroll
		for (int i = 0; i < T; ++i) {
			const unsigned long long key = ((unsigned long long)D[i] << 32) | ((unsigned long long)r << KEY_JBITS) | aj[i];
			best[i] = min(best[i], key);
		}
	}
	// Lanes whose indices differ by a multiple of 2^f hold candidates for the SAME projection entry (the last column has
	// f = 0: a million atomics on one word took 0.76 ms): reduce them inside the wave first, one atomic per entry and wave.
	const bool wave_reduce = col.f < 6u && total_threads >= 64u;  // total_threads is a power of two: every wave is full
	if (wave_reduce) {
		for (uint32_t stride = 32; stride >= (1u << col.f); stride >>= 1) {
#pragma unroll
			for (int i = 0; i < T; ++i) best[i] = min(best[i], (unsigned long long)__shfl_xor(best[i], (int)stride));
			if (stride == 1u) break;
		}
		if ((threadIdx.x & 63u) >> col.f) return;
	}
#pragma unroll
	for (int i = 0; i < T; ++i) atomicMin(&P.keys[(size_t)y * T + i], best[i]);
}

// Any pedigree the templated kernels above do not cover (three trios: T = 64; more than six individuals): one thread per
// (projection entry, chunk of ending patterns, transmission value i), everything with run-time loops.  Same keys, same
// column_finalize; no run kernel and no bit-plane records -- the slow, general path (src/pedigreedptable.cpp:240-327).
constexpr int WIDE_MAXIND = MAX_IND_WIDE;
struct WideStage {
	int32_t lut[WIDE_MAXIND][COL_CHUNKS][32];
	DevTerm terms[COL_MAXTERMS];
	uint32_t tptr[MAX_T_WIDE + 1];
};
__global__ __launch_bounds__(256) void column_step_wide(DevProblem P, uint32_t c, const uint32_t* __restrict__ prev, uint32_t total_threads) {
	const DevColumn col = P.cols[c];
	__shared__ WideStage S;
	const uint32_t T = P.T, tbits = P.tbits, n_ind = P.n_ind, k = col.k;
	{
		const int32_t* __restrict__ dl = P.delta + col.delta_off;
		for (uint32_t idx = threadIdx.x; idx < n_ind * COL_CHUNKS * 32u; idx += blockDim.x) {
			const uint32_t s = idx / (COL_CHUNKS * 32), chunk = (idx / 32) % COL_CHUNKS, v = idx & 31u;
			int32_t sum = 0;
			for (uint32_t j = 0; j < 5; ++j) {
				const uint32_t bit = chunk * 5 + j;
				if (bit < k && ((v >> j) & 1u)) sum += dl[s * k + bit];
			}
			S.lut[s][chunk][v] = sum;
		}
		const uint32_t* __restrict__ tp = P.term_ptr + col.term_off;
		// (a column with more terms than the stage holds -- nine individuals with untrusted genotypes: 3^6 per transmission value --
		// reads them from global memory)
		const uint32_t t0 = tp[0], nterms = tp[T] - t0;
		if (nterms <= (uint32_t)COL_MAXTERMS)
			for (uint32_t i = threadIdx.x; i < nterms; i += blockDim.x) S.terms[i] = P.terms[t0 + i];
		for (uint32_t i = threadIdx.x; i <= T; i += blockDim.x) S.tptr[i] = tp[i] - t0;
		__syncthreads();
	}
	const uint32_t term_count = S.tptr[T];
	const DevTerm* __restrict__ col_terms = term_count <= (uint32_t)COL_MAXTERMS ? S.terms : P.terms + P.term_ptr[col.term_off];
	const uint32_t nchunks = (k + 4) / 5;
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= total_threads) return;
	const uint32_t i = gid & (T - 1u), rest = gid >> tbits;
	const uint32_t y = rest & ((1u << col.f) - 1u), chunk = rest >> col.f;
	const uint32_t* __restrict__ segs = P.segs + col.seg_off;
	const uint32_t xbase = deposit(y, segs, col.nseg_fwd);
	const uint32_t lowmask = (1u << col.b) - 1u;
	unsigned long long best = ~0ull;
	const uint32_t ne = 1u << col.eloop;
	for (uint32_t el = 0; el < ne; ++el) {
		const uint32_t e = (chunk << col.eloop) | el;
		const uint32_t x = xbase | deposit(e, segs + col.nseg_fwd, col.nseg_end);
		int32_t L[WIDE_MAXIND];
		for (uint32_t s = 0; s < n_ind; ++s) L[s] = 0;
		for (uint32_t cc = 0; cc < nchunks; ++cc) {
			const uint32_t v = (x >> (5 * cc)) & 31u;
			for (uint32_t s = 0; s < n_ind; ++s) L[s] += S.lut[s][cc][v];
		}
		uint32_t cost = 0xFFFFFFFFu;
		for (uint32_t q = S.tptr[i]; q < S.tptr[i + 1]; ++q) {
			const DevTerm tm = col_terms[q];
			uint32_t v = tm.c;
			for (uint32_t s = 0; s < n_ind; ++s) {
				v += ((tm.plus >> s) & 1u) ? (uint32_t)L[s] : 0u;
				v -= ((tm.minus >> s) & 1u) ? (uint32_t)L[s] : 0u;
			}
			cost = min(cost, v);
		}
		uint32_t m = 0xFFFFFFFFu, mj = 0;
		if (cost != 0xFFFFFFFFu) {
			const uint32_t* __restrict__ pv = c ? prev + (size_t)(x & lowmask) * T : nullptr;
			for (uint32_t j = 0; j < T; ++j) {
				const uint32_t pj = pv ? pv[j] : 0u;
				if (pj == 0xFFFFFFFFu) continue;
				const uint32_t val = cost + pj + (uint32_t)__popc(i ^ j) * col.recomb;
				if (val < m) { m = val; mj = j; }
			}
		}
		best = min(best, ((unsigned long long)m << 32) | ((unsigned long long)gray_rank(x) << KEY_JBITS) | mj);
	}
	// lanes whose indices differ by a multiple of T * 2^f hold candidates for the same (entry, i): reduce inside the wave first
	const uint32_t span_bits = col.f + tbits;
	if (span_bits < 6u && total_threads >= 64u) {
		for (uint32_t stride = 32; stride >= (1u << span_bits); stride >>= 1) {
			best = min(best, (unsigned long long)__shfl_xor(best, (int)stride));
			if (stride == 1u) break;
		}
		if ((threadIdx.x & 63u) >> span_bits) return;
	}
	atomicMin(&P.keys[(size_t)y * T + i], best);
}

// keys -> Pr_c (value) + raw u32 backtrace record (rank << KEY_JBITS | argj); re-arms the key scratch.
__global__ __launch_bounds__(256) void column_finalize(DevProblem P, uint32_t c, uint32_t* __restrict__ cur, uint32_t entries) {
	const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= entries) return;
	const DevColumn col = P.cols[c];
	const unsigned long long key = P.keys[idx];
	P.keys[idx] = ~0ull;
	cur[idx] = (uint32_t)(key >> 32);
	reinterpret_cast<uint32_t*>(P.bt + col.bt_off)[idx] = (uint32_t)key;
	if (col.is_last) P.last_keys[idx] = key;
}
