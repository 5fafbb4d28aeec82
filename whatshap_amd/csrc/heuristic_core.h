// heuristic_core.h -- the PedMecHeuristic solver (heuristic.h), written ONCE against a small execution interface and
// instantiated twice: as a persistent single-workgroup HIP kernel (heuristic_device.hip) and single-threaded on the host
// (heuristic_host.cpp, CPU diagnostic).  The includer defines, before including this file:
//   HEUR_FN                         function qualifier (__device__ / nothing)
//   HEUR_SHARED                     storage of block-shared scratch (__shared__ / static)
//   HEUR_TID, HEUR_NT               index of the calling thread, number of threads (a power of two <= 1024)
//   HEUR_SYNC()                     barrier of all threads (also orders their global-memory accesses)
//   heur_cas32(p, cmp, val) -> old  heur_min32(p, v)  heur_min64(p, v)  heur_add32(p, v) -> old     atomics
//   heur_block_inclusive(v, tmp)    inclusive prefix sum of one value per thread over the block (tmp: 32 shared words; ends with a barrier)
//   heur_load32(p)  heur_load64(p)  loads of words other threads changed with atomics (device: past the CU's L1, which the
//                                   atomics -- performed in L2 -- do not update)
//
// Restates src/pedmecheuristic.cpp:123-409 (solve) and :420-622 (updateSolution, getRecombinationCost, getMutationCost,
// getOptPhasing, addBalance, extendSolutions, filterSolutions).  MecScore is float (src/mecheader.h): every score operation
// below is the reference's, in its order, in IEEE single precision with contraction OFF -- the beam's decisions (which
// duplicate wins, which copy is kept, where the pruning threshold falls) are then the reference's.
#pragma once
#include <cstdint>
#include <type_traits>

#pragma clang fp contract(off)

#ifndef HEUR_STAMP            // cycle stamps per phase of a column (device, WHAMD_HEURISTIC_STAMPS): stats[8 + phase] += cycles since the last stamp
#define HEUR_STAMP(D, phase)
#define HEUR_STAMP_BEGIN(D)
#endif

namespace whamd {

constexpr uint32_t HEUR_MAXS = 8;          // samples of one table
constexpr uint32_t HEUR_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t HEUR_LDS_SCRATCH = 1024;   // beams up to this size: the per-solution scratch words (flags, ranks, slots, pruning values) in LDS
constexpr uint32_t HEUR_LDS_BEAM = 1024;   // beams up to this size: hash table of the projection (2 x as many slots) and projected bipartitions in LDS

// One pool of solutions (structure of arrays); two of them are used alternately.  Every array has the SOLUTION index innermost:
// the threads of a wavefront work on consecutive solutions, so each of their loads and stores is one contiguous piece of memory
// (with the solution outermost every access of a wave touched 64 cache lines -- three quarters of the kernel's time).
struct HeurPool {
	float* score;      // [cap]
	float* mut;        // [cap] mutationScore
	uint32_t* trans;   // [cap]
	uint32_t* bt;      // [cap] btRow
	uint32_t* bits;    // [nw][cap] bipartition over the column's active reads (kept reads first, then the new ones)
	float* bal;        // [2 S][w_max][cap]
};

// Everything the kernel is handed, small enough to stay in scalar registers (as one pointer per array -- 45 of them -- the structure
// took 125 SGPRs and the compiler kept it in VGPR lanes: a v_readlane in front of most memory instructions).
struct HeurDev {
	// ---- plan (heuristic.h HeurPlan)
	uint32_t n_cols, n_samples, n_trios, tm_bits, row_limit, distrust, w_max, nw;
	uint32_t cap, tsz;             // solutions a pool holds; hash slots allocated
	const uint32_t* trios;         // [3 n_trios] sample ranks: mother-side parent, father-side parent, child (heuristic.h)
	const float* recomb; const float* mutation;
	const int8_t* genotype;
	const uint32_t* start_index;
	const HeurColMeta* col;        // [n_cols]
	const uint32_t* kept;
	const HeurReadMeta* reads;     // per starting read
	const float* new_balance; const int32_t* new_target;
	// ---- state: one block of words per pool / for the scratch arrays / for the hash table (heur_pool and the accessors below)
	uint32_t* pool_words[2];       // score | mut | trans | bt | bits [nw] | bal [2 S][w_max], each [cap]
	uint32_t* scratch;             // slot | rank | aux | val | pbits [nw], each [cap]
	uint32_t* hash;                // table [tsz] | lead [tsz] | best [tsz] (64-bit)
	// ---- records (backtrace): per column `stride` words per solution: btRow, trans, bits of the new reads
	uint32_t* arena; unsigned long long arena_words;
	unsigned long long* col_off; uint32_t* col_count;
	// ---- results
	uint8_t* opt_bipart; uint32_t* opt_trans;
	unsigned long long* stats;     // [0] status (0 ok, 1 pool overflow, 2 arena overflow), [1] widest column, [2] sum of the column sizes
};
constexpr size_t heur_pool_words(uint32_t cap, uint32_t nw, uint32_t n_samples, uint32_t w_max) { return (size_t)cap * (4u + nw + 2u * n_samples * w_max); }
constexpr size_t heur_scratch_words(uint32_t cap, uint32_t nw) { return (size_t)cap * (4u + nw); }
constexpr size_t heur_hash_words(uint32_t tsz) { return (size_t)tsz * 4u; }
HEUR_FN inline HeurPool heur_pool(const HeurDev& D, uint32_t q) {
	uint32_t* b = q ? D.pool_words[1] : D.pool_words[0];
	const size_t cap = D.cap;
	return HeurPool{reinterpret_cast<float*>(b), reinterpret_cast<float*>(b + cap), b + 2 * cap, b + 3 * cap, b + 4 * cap, reinterpret_cast<float*>(b + (4 + D.nw) * cap)};
}
HEUR_FN inline uint32_t* heur_slot(const HeurDev& D) { return D.scratch; }                                  // [cap] hash slot / side-1 mutation score
HEUR_FN inline uint32_t* heur_rank(const HeurDev& D) { return D.scratch + (size_t)D.cap; }                  // [cap] scan results
HEUR_FN inline uint32_t* heur_aux(const HeurDev& D) { return D.scratch + 2 * (size_t)D.cap; }               // [cap] flags / counts
HEUR_FN inline float* heur_val(const HeurDev& D) { return reinterpret_cast<float*>(D.scratch + 3 * (size_t)D.cap); }   // [cap] pruning values
HEUR_FN inline uint32_t* heur_pbits(const HeurDev& D) { return D.scratch + 4 * (size_t)D.cap; }             // [nw][cap] projected bipartitions
HEUR_FN inline uint32_t* heur_table(const HeurDev& D) { return D.hash; }                                    // [tsz] a member of the slot's group
HEUR_FN inline uint32_t* heur_lead(const HeurDev& D) { return D.hash + D.tsz; }                             // [tsz] smallest member index
HEUR_FN inline unsigned long long* heur_best(const HeurDev& D) { return reinterpret_cast<unsigned long long*>(D.hash + 2 * (size_t)D.tsz); }   // [tsz] min (score, index)

HEUR_FN inline uint32_t heur_sortable(float f) {
	uint32_t u = __builtin_bit_cast(uint32_t, f);
	return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
HEUR_FN inline float heur_unsortable(uint32_t k) {
	const uint32_t u = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
	return __builtin_bit_cast(float, u);
}
HEUR_FN inline float heur_abs(float x) { return x < 0 ? -x : x; }           // std::abs (sign of zero irrelevant below)
HEUR_FN inline float heur_min(float a, float b) { return b < a ? b : a; }    // std::min
HEUR_FN inline float heur_max(float a, float b) { return a < b ? b : a; }    // std::max
HEUR_FN inline uint32_t heur_popc(uint32_t v) { return (uint32_t)__builtin_popcount(v); }

// The balance matrix [2 S][w_max] of ONE solution (bal points at its element (0, 0); consecutive elements are `cap` apart) with
// one row seen as (row + add): what addBalance leaves behind, without materialising it.
struct HeurBalView {
	const float* bal; size_t cap; uint32_t w_max; uint32_t over_row; const float* add;
	HEUR_FN inline float at(uint32_t row, uint32_t i) const {
		const float v = bal[((size_t)row * w_max + i) * cap];
		return row == over_row ? v + add[i] : v;
	}
};

// getMutationCost (:438-468)
HEUR_FN inline float heur_mutation_cost(const HeurDev& D, const HeurBalView& B, uint32_t t, uint32_t p, bool allow_flips, uint32_t ahead, uint32_t w) {
	float cost = 0.0f;
	const float mc = D.mutation[p];
	const uint32_t last = ahead < w - 1u ? ahead : w - 1u;
	for (uint32_t i = 0; i <= last; ++i) {
		for (uint32_t k = 0; k < D.n_trios; ++k) {
			const uint32_t m2c = (t >> (2 * k)) & 1u, f2c = (t >> (2 * k + 1)) & 1u;
			const uint32_t t0 = D.trios[3 * k], t1 = D.trios[3 * k + 1], t2 = D.trios[3 * k + 2];
			const float cm = B.at(2 * t2, i), cf = B.at(2 * t2 + 1, i), m = B.at(2 * t0 + m2c, i), f = B.at(2 * t1 + f2c, i);
			if (allow_flips) {
				if (cm * m < 0) cost += heur_min(mc, heur_min(heur_abs(cm), heur_abs(m)));
				if (cf * f < 0) cost += heur_min(mc, heur_min(heur_abs(cf), heur_abs(f)));
			} else {
				cost += (float)(int)(cm * m < 0) * mc;
				cost += (float)(int)(cf * f < 0) * mc;
			}
		}
	}
	return cost;
}

// getOptPhasing (:471-563).  firsts[2 S]; returns the minimal cost; opt_phase[S] (0 = 0|0, 1 = 0|1, 2 = 1|0, 3 = 1|1) and
// mutated[2 S] of the first combination attaining it when the pointers are given.
HEUR_FN inline float heur_opt_phasing(const HeurDev& D, const float* firsts, uint32_t t, uint32_t p, uint8_t* opt_phase, uint8_t* mutated) {
	const uint32_t S = D.n_samples;
	float pc[HEUR_MAXS][5];
	uint8_t phases[HEUR_MAXS][4], n_ph[HEUR_MAXS], v[HEUR_MAXS];
	const float mc = D.mutation[p];
	for (uint32_t s = 0; s < S; ++s) {
		const float a0 = firsts[2 * s], a1 = firsts[2 * s + 1];
		pc[s][0] = (a0 * (float)(int)(a0 > 0) + a1 * (float)(int)(a1 > 0));
		pc[s][1] = (-a0 * (float)(int)(a0 < 0) + a1 * (float)(int)(a1 > 0));
		pc[s][2] = (a0 * (float)(int)(a0 > 0) - a1 * (float)(int)(a1 < 0));
		pc[s][3] = (-a0 * (float)(int)(a0 < 0) - a1 * (float)(int)(a1 < 0));
		float mx = pc[s][0];   // std::max_element: the first of the largest
		for (int q = 1; q < 4; ++q) if (mx < pc[s][q]) mx = pc[s][q];
		pc[s][4] = mx;
		n_ph[s] = 0;
		if (D.distrust) {
			for (uint8_t q = 0; q < 4; ++q) if (pc[s][q] < pc[s][4] + (float)2 * mc) phases[s][n_ph[s]++] = q;
		} else {
			const int8_t g = D.genotype[(size_t)s * D.n_cols + p];
			if (g == 0) phases[s][n_ph[s]++] = 0;
			else if (g == 2) phases[s][n_ph[s]++] = 3;
			else { phases[s][n_ph[s]++] = 1; phases[s][n_ph[s]++] = 2; }
		}
		v[s] = 0;
	}
	float min_cost = __builtin_inff();
	while (v[S - 1] < n_ph[S - 1]) {
		float cost = 0.0f;
		uint8_t mut[2 * HEUR_MAXS];
		for (uint32_t s = 0; s < 2 * S; ++s) mut[s] = 0;
		for (uint32_t k = 0; k < D.n_trios; ++k) {
			const uint32_t m2c = (t >> (2 * k)) & 1u, f2c = (t >> (2 * k + 1)) & 1u;
			const uint32_t t0 = D.trios[3 * k], t1 = D.trios[3 * k + 1], t2 = D.trios[3 * k + 2];
			const int acm = phases[t2][v[t2]] & 1, acf = (phases[t2][v[t2]] & 2) >> 1;
			const int am = (phases[t0][v[t0]] & (1 + m2c)) >> m2c, af = (phases[t1][v[t1]] & (1 + f2c)) >> f2c;
			cost += (float)(int)(am != acm) * mc;
			cost += (float)(int)(af != acf) * mc;
			mut[2 * t2] = am != acm;
			mut[2 * t2 + 1] = af != acf;
		}
		for (uint32_t s = 0; s < S; ++s) cost += pc[s][phases[s][v[s]]];
		if (cost < min_cost) {
			min_cost = cost;
			if (opt_phase) for (uint32_t s = 0; s < S; ++s) opt_phase[s] = phases[s][v[s]];
			if (mutated) for (uint32_t s = 0; s < 2 * S; ++s) mutated[s] = mut[s];
		}
		v[0]++;
		for (uint32_t j = 0; j + 1 < S; ++j)
			if (v[j] >= n_ph[j]) { v[j] = 0; v[j + 1]++; }
	}
	return min_cost;
}

// ---- block-wide primitives --------------------------------------------------------------------------------------------
// Exclusive prefix sums of in[0 .. n) into out (may alias in); returns the total.  Every thread must call it.
HEUR_FN inline uint32_t heur_scan(const uint32_t* in, uint32_t* out, uint32_t n) {
	HEUR_SHARED uint32_t tmp[32];
	HEUR_SHARED uint32_t carry;
	const uint32_t tid = HEUR_TID, nt = HEUR_NT;
	if (tid == 0) carry = 0;
	HEUR_SYNC();
	for (uint32_t base = 0; base < n; base += nt) {
		const uint32_t i = base + tid;
		const uint32_t mine = i < n ? in[i] : 0u;
		const uint32_t incl = heur_block_inclusive(mine, tmp);
		const uint32_t c = carry;
		if (i < n) out[i] = c + incl - mine;
		HEUR_SYNC();
		if (tid == nt - 1) carry = c + incl;
		HEUR_SYNC();
	}
	return carry;
}

// The k-th smallest (0-based) of the sortable keys of val[0 .. n): radix select, four 8-bit passes; the bucket of a pass is found
// with a block-wide prefix sum over the 256 counters.
HEUR_FN inline uint32_t heur_select(const float* val, uint32_t n, uint32_t k) {
	HEUR_SHARED uint32_t hist[256];
	HEUR_SHARED uint32_t tmp[32];
	HEUR_SHARED uint32_t sh_prefix, sh_k;
	const uint32_t tid = HEUR_TID, nt = HEUR_NT;
	uint32_t prefix = 0, mask = 0, kk = k;
	for (int pass = 3; pass >= 0; --pass) {
		for (uint32_t b = tid; b < 256u; b += nt) hist[b] = 0;
		HEUR_SYNC();
		for (uint32_t i = tid; i < n; i += nt) {
			const uint32_t key = heur_sortable(val[i]);
			if ((key & mask) == prefix) heur_add32(&hist[(key >> (8 * pass)) & 255u], 1u);
		}
		HEUR_SYNC();
		// bucket b with  sum(hist[0 .. b)) <= kk < sum(hist[0 .. b])
		uint32_t carry = 0;
		for (uint32_t base = 0; base < 256u; base += nt) {   // (one round when the block has >= 256 threads)
			const uint32_t b = base + tid;
			const uint32_t mine = b < 256u ? hist[b] : 0u;
			const uint32_t incl = carry + heur_block_inclusive(mine, tmp);
			if (b < 256u && incl - mine <= kk && kk < incl) { sh_prefix = prefix | (b << (8 * pass)); sh_k = kk - (incl - mine); }
			HEUR_SYNC();
			if (nt < 256u) { if (tid == nt - 1) tmp[0] = incl; HEUR_SYNC(); carry = tmp[0]; HEUR_SYNC(); }
		}
		prefix = sh_prefix;
		kk = sh_k;
		mask |= 0xFFu << (8 * pass);
		HEUR_SYNC();
	}
	return prefix;
}

// Copies / updates of one row of a balance matrix (elements `st` apart), eight elements at a time: all loads of a batch are issued
// before the first store (source and destination may be the same pool, which the compiler must assume to alias -- element by element
// every load would wait for the previous store, a full memory round trip per element).
constexpr uint32_t HEUR_BATCH = 8;
// dst[x] = x + shift < n_src ? src[x + shift] : 0   for x < w
HEUR_FN inline void heur_copy_row(float* dst, const float* src, size_t st, uint32_t w, uint32_t shift, uint32_t n_src) {
	uint32_t x0 = 0;
	const uint32_t n_full = w < n_src - (n_src < shift ? n_src : shift) ? w : n_src - (n_src < shift ? n_src : shift);   // positions with a source
	for (; x0 + HEUR_BATCH <= n_full; x0 += HEUR_BATCH) {   // whole batches inside the source: no per-element tests
		float t[HEUR_BATCH];
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) t[u] = src[(size_t)(x0 + u + shift) * st];
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) dst[(size_t)(x0 + u) * st] = t[u];
	}
	// the last, partial batch: UNCONDITIONAL loads (element 0 of the row where there is nothing to fetch, the value dropped afterwards).  Written as
	// `cond ? src[...] : 0` every element was its own branch with a wait behind the load -- up to seven memory round trips in a row, in a kernel
	// whose one workgroup per CU has nothing else to run meanwhile.
	for (; x0 < w; x0 += HEUR_BATCH) {
		float t[HEUR_BATCH];
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) {
			const bool has = x0 + u < w && x0 + u + shift < n_src;
			t[u] = src[(size_t)(has ? x0 + u + shift : 0u) * st];
		}
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) {
			const bool has = x0 + u < w && x0 + u + shift < n_src;
			if (x0 + u < w) dst[(size_t)(x0 + u) * st] = has ? t[u] : 0.0f;
		}
	}
}
// the same for the two rows of one sample at once (rows come in pairs: haplotype 0 / 1): sixteen loads in flight per batch
HEUR_FN inline void heur_copy_row_pair(float* dst, const float* src, size_t st, size_t row_st, uint32_t w, uint32_t shift, uint32_t n_src) {
	uint32_t x0 = 0;
	const uint32_t n_full = w < n_src - (n_src < shift ? n_src : shift) ? w : n_src - (n_src < shift ? n_src : shift);
	for (; x0 + HEUR_BATCH <= n_full; x0 += HEUR_BATCH) {
		float t[HEUR_BATCH], u2[HEUR_BATCH];
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) { t[u] = src[(size_t)(x0 + u + shift) * st]; u2[u] = src[row_st + (size_t)(x0 + u + shift) * st]; }
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) { dst[(size_t)(x0 + u) * st] = t[u]; dst[row_st + (size_t)(x0 + u) * st] = u2[u]; }
	}
	if (x0 < w) {
		heur_copy_row(dst + (size_t)x0 * st, src + (size_t)x0 * st, st, w - x0, shift, n_src > x0 ? n_src - x0 : 0u);
		heur_copy_row(dst + row_st + (size_t)x0 * st, src + row_st + (size_t)x0 * st, st, w - x0, shift, n_src > x0 ? n_src - x0 : 0u);
	}
}
// row[x] += add[x]   for x < w
HEUR_FN inline void heur_add_row(float* row, size_t st, const float* add, uint32_t w) {
	uint32_t x0 = 0;
	for (; x0 + HEUR_BATCH <= w; x0 += HEUR_BATCH) {
		float t[HEUR_BATCH];
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) t[u] = row[(size_t)(x0 + u) * st];
		float a[HEUR_BATCH];
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) a[u] = add[x0 + u];
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) row[(size_t)(x0 + u) * st] = t[u] + a[u];
	}
	for (; x0 < w; x0 += HEUR_BATCH) {   // (the partial batch: unconditional loads from clamped positions, see heur_copy_row)
		float t[HEUR_BATCH], a[HEUR_BATCH];
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) {
			const uint32_t x = x0 + u < w ? x0 + u : x0;   // (x0 < w: a valid position)
			t[u] = row[(size_t)x * st];
			a[u] = add[x];
		}
#pragma unroll
		for (uint32_t u = 0; u < HEUR_BATCH; ++u) if (x0 + u < w) row[(size_t)(x0 + u) * st] = t[u] + a[u];
	}
}

HEUR_FN inline void heur_copy_solution(const HeurDev& D, const HeurPool& src, uint32_t i, const HeurPool& dst, uint32_t j, uint32_t w) {
	const float sc = src.score[i], mu = src.mut[i];
	const uint32_t tr = src.trans[i], bt = src.bt[i];
	dst.score[j] = sc; dst.mut[j] = mu; dst.trans[j] = tr; dst.bt[j] = bt;
	const size_t cap = D.cap;
	for (uint32_t q = 0; q < D.nw; ++q) dst.bits[q * cap + j] = src.bits[q * cap + i];
	const uint32_t rows = 2u * D.n_samples;
	for (uint32_t r = 0; r < rows; r += 2) heur_copy_row_pair(dst.bal + (size_t)r * D.w_max * cap + j, src.bal + (size_t)r * D.w_max * cap + i, cap, (size_t)D.w_max * cap, w, 0, w);
}

// filterSolutions (:604-622): pool[cur] (count) -> pool[cur ^ 1]; returns the new count.
HEUR_FN inline uint32_t heur_filter(const HeurDev& D, uint32_t cur, uint32_t count, uint32_t w, float* val, uint32_t* aux, uint32_t* rank) {
	const uint32_t tid = HEUR_TID, nt = HEUR_NT;
	const HeurPool src = heur_pool(D, cur);
	const HeurPool dst = heur_pool(D, cur ^ 1u);
	for (uint32_t i = tid; i < count; i += nt) val[i] = src.score[i] + src.mut[i];
	HEUR_SYNC();
	HEUR_SHARED uint32_t sh_low;
	if (tid == 0) sh_low = 0xFFFFFFFFu;
	HEUR_SYNC();
	{
		uint32_t low = 0xFFFFFFFFu;
		for (uint32_t i = tid; i < count; i += nt) { const uint32_t key = heur_sortable(val[i]); low = key < low ? key : low; }
		heur_min32(&sh_low, low);
	}
	HEUR_SYNC();
	const float lowest = heur_unsortable(sh_low);
	const float too_high = count > D.row_limit ? heur_unsortable(heur_select(val, count, D.row_limit)) : __builtin_inff();
	for (uint32_t i = tid; i < count; i += nt) aux[i] = (val[i] < too_high || val[i] == lowest) ? 1u : 0u;
	HEUR_SYNC();
	uint32_t kept = heur_scan(aux, rank, count);
	if (kept > HEUR_MAX_ROW_LIMIT) kept = HEUR_MAX_ROW_LIMIT;   // `kept.size() < MAX_ROW_LIMIT`: the first 65535 in order
	for (uint32_t i = tid; i < count; i += nt)
		if (aux[i] && rank[i] < kept) heur_copy_solution(D, src, i, dst, rank[i], w);
	HEUR_SYNC();
	return kept;
}

// bit b of solution i's bipartition (bits: [nw][cap])
HEUR_FN inline uint32_t heur_get_bit(const uint32_t* bits, size_t cap, uint32_t i, uint32_t b) { return (bits[(b >> 5) * cap + i] >> (b & 31u)) & 1u; }

// ---- the solver: src/pedmecheuristic.cpp:123-358 ------------------------------------------------------------------------
HEUR_FN inline void heur_solve(const HeurDev& D) {
	const uint32_t tid = HEUR_TID, nt = HEUR_NT;
	const uint32_t S = D.n_samples, rows = 2u * S, nw = D.nw, wm = D.w_max, T = 1u << D.tm_bits;
	uint32_t cur = 0, count = 1, w_prev = 1;
	unsigned long long arena_used = 0, widest = 0, total = 0;
	// lastCol = { empty bipartition, transmission 0, score 0, balances (1, 0) }  (:151)
	const HeurPool first = heur_pool(D, 0);
	if (tid == 0) { first.score[0] = 0.0f; first.mut[0] = 0.0f; first.trans[0] = 0; first.bt[0] = 0; }
	const size_t cap = D.cap;
	// per-solution scratch of the phases: in LDS while the beam is small (every phase between two barriers otherwise starts with a
	// round trip to L2 for a word the same thread wrote a phase earlier).  The stages below are generic lambdas instantiated for the
	// LDS arrays and for the global ones -- pointers chosen at run time would turn every access into a FLAT instruction.
	HEUR_SHARED uint32_t sh_aux[HEUR_LDS_SCRATCH], sh_rank[HEUR_LDS_SCRATCH], sh_slot[HEUR_LDS_SCRATCH];
	HEUR_SHARED float sh_val[HEUR_LDS_SCRATCH];
	auto filter = [&](uint32_t pool, uint32_t n, uint32_t w) -> uint32_t {
		if (n <= HEUR_LDS_SCRATCH) return heur_filter(D, pool, n, w, sh_val, sh_aux, sh_rank);
		return heur_filter(D, pool, n, w, heur_val(D), heur_aux(D), heur_rank(D));
	};
	for (uint32_t x = tid; x < rows * wm; x += nt) first.bal[x * cap] = 0.0f;
	for (uint32_t x = tid; x < nw; x += nt) first.bits[x * cap] = 0;
	HEUR_SYNC();
	HEUR_STAMP_BEGIN(D);
	for (uint32_t p = 0; p < D.n_cols; ++p) {
		const HeurColMeta cm = D.col[p];
		const uint32_t w = cm.window, nk = cm.n_kept, nn = cm.n_new;
		const uint32_t* kept = D.kept + cm.kept_off;
		// ================= projection onto the reads that continue, duplicates merged into their first occurrence (:170-197)
		{
			const HeurPool src = heur_pool(D, cur);
			const HeurPool dst = heur_pool(D, cur ^ 1u);
			uint32_t tsz = 64;
			while (tsz < 2u * count) tsz <<= 1;
			// the usual beam: hash table and projected bipartitions in LDS (its atomics do not leave the CU)
			HEUR_SHARED uint32_t sh_table[2 * HEUR_LDS_BEAM], sh_lead[2 * HEUR_LDS_BEAM], sh_pbits[2 * HEUR_LDS_BEAM];
			HEUR_SHARED unsigned long long sh_best[2 * HEUR_LDS_BEAM];
			HEUR_SHARED uint32_t sh_kept[64], sh_trans[HEUR_LDS_BEAM];
			const bool in_lds = count <= HEUR_LDS_BEAM && nw <= 2u;
			if (nw <= 2u) for (uint32_t a = tid; a < nk; a += nt) sh_kept[a] = kept[a];   // (read by every thread below: past the first barrier)
			// (a generic lambda, instantiated for the LDS arrays and for the global ones: with pointers chosen at run time every access
			// would be a FLAT instruction -- measured: the probe loop alone took a third of the column)
			uint32_t n2 = 0;
			auto project = [&](uint32_t* table, uint32_t* lead, unsigned long long* best, uint32_t* pbits, uint32_t* trans_stage, const uint32_t* ptrans, const size_t pst,
			                   uint32_t* aux, uint32_t* rank, uint32_t* slot) {
				for (uint32_t x = tid; x < tsz; x += nt) { table[x] = HEUR_EMPTY; lead[x] = HEUR_EMPTY; best[x] = ~0ull; }
				HEUR_SYNC();
				HEUR_STAMP(D, 7);
				for (uint32_t i = tid; i < count; i += nt) {
					if (trans_stage) trans_stage[i] = src.trans[i];
					if (nw <= 2u) {   // both words in registers: bit a of the projection = bit kept[a] of the solution
						const unsigned long long sb = (unsigned long long)src.bits[i] | (nw > 1u ? (unsigned long long)src.bits[cap + i] << 32 : 0ull);
						unsigned long long pb = 0;
						for (uint32_t a = 0; a < nk; ++a) pb |= ((sb >> sh_kept[a]) & 1ull) << a;
						pbits[i] = (uint32_t)pb;
						if (nw > 1u) pbits[pst + i] = (uint32_t)(pb >> 32);
					} else {
						for (uint32_t q = 0; q < nw; ++q) pbits[q * pst + i] = 0;
						for (uint32_t a = 0; a < nk; ++a) pbits[(a >> 5) * pst + i] |= heur_get_bit(src.bits, cap, i, kept[a]) << (a & 31u);
					}
				}
				HEUR_SYNC();
				HEUR_STAMP(D, 8);
				for (uint32_t i = tid; i < count; i += nt) {
					const uint32_t tr = ptrans[i];
					const unsigned long long score_key = ((unsigned long long)heur_sortable(src.score[i]) << 32) | i;
					uint32_t h = tr * 0x9E3779B1u + 0x7F4A7C15u;
					for (uint32_t q = 0; q < nw; ++q) { h ^= pbits[q * pst + i]; h *= 0x85EBCA6Bu; h ^= h >> 13; }
					h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;   // full avalanche: the newest reads are the HIGH bits of the bipartition
					uint32_t pos = h & (tsz - 1u);
					for (;;) {
						uint32_t other = heur_load32(&table[pos]);
						if (other == HEUR_EMPTY) {
							other = heur_cas32(&table[pos], HEUR_EMPTY, i);
							if (other == HEUR_EMPTY) break;   // this solution is the slot's group from now on
						}
						bool same = ptrans[other] == tr;
						for (uint32_t q = 0; q < nw && same; ++q) same = pbits[q * pst + other] == pbits[q * pst + i];
						if (same) break;
						pos = (pos + 1u) & (tsz - 1u);
					}
					slot[i] = pos;
					heur_min32(&lead[pos], i);
					// updateSolution (:420-432): a later duplicate replaces the kept one only if strictly better
					heur_min64(&best[pos], score_key);
				}
				HEUR_SYNC();
				HEUR_STAMP(D, 0);
				for (uint32_t i = tid; i < count; i += nt) aux[i] = heur_load32(&lead[slot[i]]) == i ? 1u : 0u;
				HEUR_SYNC();
				n2 = heur_scan(aux, rank, count);
				for (uint32_t i = tid; i < count; i += nt) {
					if (!aux[i]) continue;
					const uint32_t j = rank[i], win = (uint32_t)heur_load64(&best[slot[i]]);
					dst.score[j] = src.score[win]; dst.mut[j] = 0.0f; dst.trans[j] = src.trans[i]; dst.bt[j] = win;
					for (uint32_t q = 0; q < nw; ++q) dst.bits[q * cap + j] = pbits[q * pst + i];
					// balances of the winner without their first position, extended with zeros to the column's window (:204-206, :425-431)
					for (uint32_t r = 0; r < rows; r += 2) heur_copy_row_pair(dst.bal + (size_t)r * wm * cap + j, src.bal + (size_t)r * wm * cap + win, cap, (size_t)wm * cap, w, 1, w_prev);
				}
			};
			if (in_lds && count <= HEUR_LDS_SCRATCH) project(sh_table, sh_lead, sh_best, sh_pbits, sh_trans, sh_trans, (size_t)HEUR_LDS_BEAM, sh_aux, sh_rank, sh_slot);
			else if (in_lds) project(sh_table, sh_lead, sh_best, sh_pbits, sh_trans, sh_trans, (size_t)HEUR_LDS_BEAM, heur_aux(D), heur_rank(D), heur_slot(D));
			else project(heur_table(D), heur_lead(D), heur_best(D), heur_pbits(D), nullptr, src.trans, cap, heur_aux(D), heur_rank(D), heur_slot(D));
			HEUR_SYNC();
			HEUR_STAMP(D, 1);
			cur ^= 1u;
			count = n2;
		}
		// ================= the reads that start here, one after the other (:240-297)
		for (uint32_t q = 0; q < nn; ++q) {
			const uint32_t nr = cm.new_off + q;
			const HeurReadMeta rm = D.reads[nr];
			const int32_t eq = rm.equal_to;
			const uint32_t bitpos = nk + q;
			const HeurPool P = heur_pool(D, cur);
			if (eq >= 0) {   // identical to an earlier read of the column: same side, no branching (:247-250)
				for (uint32_t i = tid; i < count; i += nt) {
					if (heur_get_bit(P.bits, cap, i, nk + (uint32_t)eq)) P.bits[(bitpos >> 5) * cap + i] |= 1u << (bitpos & 31u);
				}
				HEUR_SYNC();
				continue;
			}
			const uint32_t s = rm.sample;
			const bool seen = rm.seen != 0;
			const float* add = D.new_balance + rm.bal_off;
			const int32_t* target = D.new_target + rm.bal_off;   // genotype of sample s at the window's positions
			if ((unsigned long long)count * 2ull > D.cap) { if (tid == 0) D.stats[0] = 1; return; }
			uint32_t n_app = 0;
			auto place = [&](uint32_t* aux, uint32_t* rank, uint32_t* slot, float* val) {
				// pass 1: both placements of the read scored per solution; aux = 0 keep side 0, 1 keep side 1, 2 keep both
				for (uint32_t i = tid; i < count; i += nt) {
					const float* bal = P.bal + i;
					const float* b0 = bal + (size_t)(2 * s) * wm * cap;
					const float* b1 = bal + (size_t)(2 * s + 1) * wm * cap;
					// addBalance (src/pedmecheuristic.cpp:566-586; the penalty only, the rows are updated in pass 2) of the read on either
					// haplotype, and the "useful" test of the untrusted-genotype mode (:256-258), in ONE pass over the two rows, eight positions
					// at a time (all loads of a batch before the arithmetic that waits for them); each penalty is accumulated in the reference's order
					bool useful = D.distrust ? false : rm.useful != 0;
					float pen0 = 0, pen1 = 0;
					auto batch = [&](const uint32_t x0, auto whole) {   // whole batches carry no per-position tests
						constexpr bool WHOLE = decltype(whole)::value;
						// (the read's balance and the genotype of the position are loaded with the batch as well: fetched where they are used,
						// each was one more dependent round trip per position)
						float v0[HEUR_BATCH], v1[HEUR_BATCH], av[HEUR_BATCH];
						int32_t tv[HEUR_BATCH];
#pragma unroll
						for (uint32_t u = 0; u < HEUR_BATCH; ++u) {
							v0[u] = (WHOLE || x0 + u < w) ? b0[(size_t)(x0 + u) * cap] : 0.0f;
							v1[u] = (WHOLE || x0 + u < w) ? b1[(size_t)(x0 + u) * cap] : 0.0f;
							av[u] = (WHOLE || x0 + u < w) ? add[x0 + u] : 0.0f;
							tv[u] = (WHOLE || x0 + u < w) ? target[x0 + u] : 0;
						}
#pragma unroll
						for (uint32_t u = 0; u < HEUR_BATCH; ++u) {
							if (!WHOLE && x0 + u >= w) continue;
							const float a = av[u], s0 = v0[u], s1 = v1[u];
							if (D.distrust) {
								useful = useful || (a != 0 && s0 * s1 < 0) || ((a + s0) * s0 <= 0 && (a + s1) * s1 <= 0);
								if (s0 * a < 0) pen0 += heur_min(heur_abs(s0), heur_abs(a));
								if (s1 * a < 0) pen1 += heur_min(heur_abs(s1), heur_abs(a));
							} else if (tv[u] == 1) {
								if (a <= 0) { pen0 += heur_min(-a, heur_max(s0 - s1, (float)0)); pen1 += heur_min(-a, heur_max(s1 - s0, (float)0)); }
								else { pen0 += heur_min(a, heur_max(s1 - s0, (float)0)); pen1 += heur_min(a, heur_max(s0 - s1, (float)0)); }
							} else {
								const float t = heur_abs(a) * (float)(int)(a * (float)(tv[u] - 1) < 0);
								pen0 += t;
								pen1 += t;
							}
						}
					};
					uint32_t xb = 0;
					for (; xb + HEUR_BATCH <= w; xb += HEUR_BATCH) batch(xb, std::true_type{});
					if (xb < w) batch(xb, std::false_type{});
					const uint32_t tr = P.trans[i];
					const float sc = P.score[i];
					float sc1 = 0, mu1 = 0;
					if (seen) {
						sc1 = sc + pen1;
						mu1 = heur_mutation_cost(D, HeurBalView{bal, cap, wm, 2 * s + 1, add}, tr, p, true, 5, w);
					}
					const float sc0 = sc + pen0;
					const float mu0 = heur_mutation_cost(D, HeurBalView{bal, cap, wm, 2 * s, add}, tr, p, true, 5, w);
					uint32_t mode = 0;
					if (seen) mode = useful ? 2u : ((sc0 + mu0 > sc1 + mu1) ? 1u : 0u);
					aux[i] = mode;
					// (scores of both sides kept for pass 2: val = side 1's score, rank slot reused below for its mutation score)
					val[i] = sc1;
					slot[i] = __builtin_bit_cast(uint32_t, mu1);
					P.score[i] = mode == 1u ? sc1 : sc0;
					P.mut[i] = mode == 1u ? mu1 : mu0;
				}
				HEUR_SYNC();
				HEUR_STAMP(D, 2);
				for (uint32_t i = tid; i < count; i += nt) rank[i] = aux[i] == 2u ? 1u : 0u;
				HEUR_SYNC();
				n_app = heur_scan(rank, rank, count);
				// pass 2: the copies (side 1) behind the existing solutions in their order, then the read joins its side in place
				for (uint32_t i = tid; i < count; i += nt) {
					if (aux[i] != 2u) continue;
					const uint32_t j = count + rank[i];
					heur_copy_solution(D, P, i, P, j, w);
					P.score[j] = val[i];
					P.mut[j] = __builtin_bit_cast(float, slot[i]);
					heur_add_row(P.bal + (size_t)(2 * s + 1) * wm * cap + j, cap, add, w);
					P.bits[(bitpos >> 5) * cap + j] |= 1u << (bitpos & 31u);
				}
				HEUR_SYNC();
				for (uint32_t i = tid; i < count; i += nt) {
					const uint32_t side = aux[i] == 1u ? 1u : 0u;
					heur_add_row(P.bal + (size_t)(2 * s + side) * wm * cap + i, cap, add, w);
					if (side) P.bits[(bitpos >> 5) * cap + i] |= 1u << (bitpos & 31u);
				}
				HEUR_SYNC();
			};
			if (count <= HEUR_LDS_SCRATCH) place(sh_aux, sh_rank, sh_slot, sh_val);
			else place(heur_aux(D), heur_rank(D), heur_slot(D), heur_val(D));
			count += n_app;
			HEUR_STAMP(D, 3);
			if (count > D.row_limit) { count = filter(cur, count, w); cur ^= 1u; }
			HEUR_STAMP(D, 4);
		}
		// ================= other transmission values where they pay for themselves (:299-303, :588-602)
		{
			const HeurPool P = heur_pool(D, cur);
			const float rc1 = D.recomb[p];
			uint32_t n_app = 0;
			bool overflow = false;
			// getMutationCost of a transmission value without flips looks at the FIRST position only (:438-468 with ahead = 0): per trio the
			// child's two balances and both haplotypes of either parent -- six words, loaded once per solution and kept in registers for
			// every transmission value tried (fetched inside heur_mutation_cost they were four dependent round trips per value and trio)
			constexpr uint32_t TRIO_CACHE = 3;
			const bool cached = D.n_trios <= TRIO_CACHE;
			struct TrioFirsts { float v[TRIO_CACHE][6]; };
			auto load_firsts = [&](uint32_t i, TrioFirsts& c) {
#pragma unroll
				for (uint32_t k = 0; k < TRIO_CACHE; ++k) {
					if (k >= D.n_trios) continue;
					const uint32_t t0 = D.trios[3 * k], t1 = D.trios[3 * k + 1], t2 = D.trios[3 * k + 2];
					const float* b = P.bal + i;
					c.v[k][0] = b[(size_t)(2 * t2) * wm * cap]; c.v[k][1] = b[(size_t)(2 * t2 + 1) * wm * cap];
					c.v[k][2] = b[(size_t)(2 * t0) * wm * cap]; c.v[k][3] = b[(size_t)(2 * t0 + 1) * wm * cap];
					c.v[k][4] = b[(size_t)(2 * t1) * wm * cap]; c.v[k][5] = b[(size_t)(2 * t1 + 1) * wm * cap];
				}
			};
			const float mc1 = D.mutation[p];
			auto cost_of = [&](uint32_t i, const TrioFirsts& c, uint32_t t) -> float {
				if (!cached) return heur_mutation_cost(D, HeurBalView{P.bal + i, cap, wm, HEUR_EMPTY, nullptr}, t, p, false, 0, w);
				float cost = 0.0f;
#pragma unroll
				for (uint32_t k = 0; k < TRIO_CACHE; ++k) {
					if (k >= D.n_trios) continue;
					const uint32_t m2c = (t >> (2 * k)) & 1u, f2c = (t >> (2 * k + 1)) & 1u;
					const float cm = c.v[k][0], cf = c.v[k][1], m = m2c ? c.v[k][3] : c.v[k][2], f = f2c ? c.v[k][5] : c.v[k][4];
					cost += (float)(int)(cm * m < 0) * mc1;
					cost += (float)(int)(cf * f < 0) * mc1;
				}
				return cost;
			};
			auto transmit = [&](uint32_t* aux, uint32_t* rank) {
				for (uint32_t i = tid; i < count; i += nt) {
					TrioFirsts c;
					if (cached) load_firsts(i, c);
					const uint32_t tr = P.trans[i];
					const float mu = cost_of(i, c, tr);
					P.mut[i] = mu;
					uint32_t n_ext = 0;
					if (mu > 0) {
						for (uint32_t t = 0; t < T; ++t) {
							if (t == tr) continue;
							const float rc = rc1 * (float)heur_popc(tr ^ t);
							if (rc >= mu) continue;
							const float m2 = cost_of(i, c, t);
							if (m2 + rc >= mu) continue;
							++n_ext;
						}
					}
					aux[i] = n_ext;
				}
				HEUR_SYNC();
				n_app = heur_scan(aux, rank, count);
				if ((unsigned long long)count + n_app > D.cap) { if (tid == 0) D.stats[0] = 1; overflow = true; return; }
				for (uint32_t i = tid; i < count; i += nt) {
					if (!aux[i]) continue;
					TrioFirsts c;
					if (cached) load_firsts(i, c);
					const uint32_t tr = P.trans[i];
					const float mu = P.mut[i], sc = P.score[i];
					uint32_t j = count + rank[i];
					for (uint32_t t = 0; t < T; ++t) {
						if (t == tr) continue;
						const float rc = rc1 * (float)heur_popc(tr ^ t);
						if (rc >= mu) continue;
						const float m2 = cost_of(i, c, t);
						if (m2 + rc >= mu) continue;
						heur_copy_solution(D, P, i, P, j, w);
						P.trans[j] = t; P.score[j] = sc + rc; P.mut[j] = m2;
						++j;
					}
				}
				HEUR_SYNC();
			};
			if (count <= HEUR_LDS_SCRATCH) transmit(sh_aux, sh_rank);
			else transmit(heur_aux(D), heur_rank(D));
			if (overflow) return;
			count += n_app;
			if (count > D.row_limit) { count = filter(cur, count, w); cur ^= 1u; }
			HEUR_STAMP(D, 5);
		}
		// ================= the column's own phasing cost (:306-313), then the backtrace record of the column (:315-330)
		{
			const HeurPool P = heur_pool(D, cur);
			const uint32_t nwn = (nn + 31u) >> 5, stride = 2u + nwn;
			if (arena_used + (unsigned long long)count * stride > D.arena_words) { if (tid == 0) D.stats[0] = 2; return; }
			uint32_t* rec = D.arena + arena_used;
			for (uint32_t i = tid; i < count; i += nt) {
				float firsts[2 * HEUR_MAXS];
				for (uint32_t r = 0; r < rows; ++r) firsts[r] = P.bal[(size_t)r * wm * cap + i];
				P.score[i] += heur_opt_phasing(D, firsts, P.trans[i], p, nullptr, nullptr);
				uint32_t* e = rec + (size_t)i * stride;
				e[0] = P.bt[i];
				e[1] = P.trans[i];
				for (uint32_t q = 0; q < nwn; ++q) e[2 + q] = 0;
				for (uint32_t q = 0; q < nn; ++q) e[2 + (q >> 5)] |= heur_get_bit(P.bits, cap, i, nk + q) << (q & 31u);
			}
			if (tid == 0) { D.col_off[p] = arena_used; D.col_count[p] = count; }
			arena_used += (unsigned long long)count * stride;
			if (count > widest) widest = count;
			total += count;
			HEUR_SYNC();
			HEUR_STAMP(D, 6);
		}
		w_prev = w;
	}
	// ================= best solution of the last column: the first with the smallest score (:332-341), then the walk back (:343-359)
	{
		const HeurPool P = heur_pool(D, cur);
		unsigned long long* best = heur_best(D);
		if (tid == 0) best[0] = ~0ull;
		HEUR_SYNC();
		for (uint32_t i = tid; i < count; i += nt) heur_min64(&best[0], ((unsigned long long)heur_sortable(P.score[i]) << 32) | i);
		HEUR_SYNC();
		if (tid == 0) {
			uint32_t ri = (uint32_t)heur_load64(&best[0]);
			for (uint32_t p = D.n_cols; p-- > 0;) {
				const uint32_t nn = D.col[p].n_new, nwn = (nn + 31u) >> 5, stride = 2u + nwn;
				const uint32_t* e = D.arena + D.col_off[p] + (size_t)ri * stride;
				for (uint32_t q = 0; q < nn; ++q) D.opt_bipart[D.start_index[p] + q] = (uint8_t)((e[2 + (q >> 5)] >> (q & 31u)) & 1u);
				D.opt_trans[p] = e[1];
				ri = e[0];
			}
			D.stats[1] = widest;
			D.stats[2] = total;
		}
		HEUR_SYNC();
	}
}

}  // namespace whamd
