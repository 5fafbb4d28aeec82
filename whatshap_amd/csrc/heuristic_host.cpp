// heuristic_host.cpp -- heuristic_core.h instantiated for the host with ONE thread (barriers and atomics are plain statements).
// Product part: heuristic_finish -- the allele votes and the per-column optimal phasing that follow the device's beam search
// (src/pedmecheuristic.cpp:361-406 run on the host from the downloaded bipartition).  Debug part (WHAMD_DEBUG_BUILD, libwhatshap_amd_debug.so
// only): heuristic_solve_host, the whole beam search on one CPU thread, so that the CPU test-suite can check the restated solver against
// the compiled reference without a GPU (whamd_debug_pedmec_heuristic_create_host).  The PedMecHeuristic drop-in runs heuristic_device.hip
// and fails loudly without a device.
#include <algorithm>
#include <cstring>

#include "heuristic.h"

#define HEUR_FN
#define HEUR_SHARED static thread_local
#define HEUR_TID 0u
#define HEUR_NT 1u
#define HEUR_SYNC() ((void)0)
namespace whamd {
static inline uint32_t heur_cas32(uint32_t* p, uint32_t cmp, uint32_t val) { const uint32_t old = *p; if (old == cmp) *p = val; return old; }
static inline void heur_min32(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
static inline void heur_min64(unsigned long long* p, unsigned long long v) { if (v < *p) *p = v; }
static inline uint32_t heur_add32(uint32_t* p, uint32_t v) { const uint32_t old = *p; *p = old + v; return old; }
static inline uint32_t heur_block_inclusive(uint32_t v, uint32_t*) { return v; }   // one thread: its own value
static inline uint32_t heur_load32(const uint32_t* p) { return *p; }
static inline unsigned long long heur_load64(const unsigned long long* p) { return *p; }
}  // namespace whamd
#include "heuristic_core.h"

#ifdef WHAMD_DEBUG_BUILD
namespace whamd {

whamd_status_t heuristic_solve_host(const HeurPlan& pl, HeurResult& out, std::string& msg) {
	out = HeurResult();
	out.bipartition.assign(pl.n_reads, 0);
	out.transmission.assign(pl.n_cols, 0);
	if (pl.n_cols == 0) return WHAMD_OK;
	if (pl.n_samples > HEUR_MAXS || pl.n_trios > HEUR_MAXS) { msg = "too many samples for the heuristic solver"; return WHAMD_ERR_UNSUPPORTED; }
	// small inputs only: capacity for a beam that never prunes below 2^12 solutions
	const uint32_t T = 1u << pl.tm_bits;
	const uint32_t cap = std::min<uint32_t>(HEUR_MAX_ROW_LIMIT, std::max<uint32_t>(pl.row_limit, 4096u)) * std::max(2u, T);
	HeurDev D{};
	D.n_cols = pl.n_cols; D.n_samples = pl.n_samples; D.n_trios = pl.n_trios; D.tm_bits = pl.tm_bits; D.row_limit = pl.row_limit;
	D.distrust = pl.distrust; D.w_max = pl.w_max; D.nw = pl.nw;
	uint32_t tsz = 64;
	while (tsz < 2u * cap) tsz <<= 1;
	D.cap = cap; D.tsz = tsz;
	std::vector<uint32_t> pool0(heur_pool_words(cap, pl.nw, pl.n_samples, pl.w_max), 0), pool1(pool0.size(), 0), scratch(heur_scratch_words(cap, pl.nw), 0);
	std::vector<unsigned long long> hash(heur_hash_words(tsz) / 2 + 1, 0);   // (64-bit elements: the `best` part is aligned)
	D.pool_words[0] = pool0.data(); D.pool_words[1] = pool1.data(); D.scratch = scratch.data(); D.hash = reinterpret_cast<uint32_t*>(hash.data());
	unsigned long long arena_words = 0;
	for (uint32_t p = 0; p < pl.n_cols; ++p) arena_words += (unsigned long long)(2 + ((pl.n_new[p] + 31) >> 5));
	arena_words *= cap;
	arena_words = std::min<unsigned long long>(arena_words, 1ull << 28);
	std::vector<uint32_t> arena(arena_words), col_count(pl.n_cols);
	std::vector<unsigned long long> col_off(pl.n_cols), stats(4, 0);
	const std::vector<HeurColMeta> col_meta = heuristic_col_meta(pl);
	const std::vector<HeurReadMeta> read_meta = heuristic_read_meta(pl);
	D.trios = pl.trios.data();
	D.recomb = pl.recomb.data(); D.mutation = pl.mutation.data(); D.genotype = pl.genotype.data(); D.start_index = pl.start_index.data();
	D.col = col_meta.data(); D.kept = pl.kept.data(); D.reads = read_meta.data();
	D.new_balance = pl.new_balance.data(); D.new_target = pl.new_target.data();
	D.arena = arena.data(); D.arena_words = arena_words; D.col_off = col_off.data(); D.col_count = col_count.data();
	D.opt_bipart = out.bipartition.data(); D.opt_trans = out.transmission.data(); D.stats = stats.data();
	heur_solve(D);
	if (stats[0]) { msg = stats[0] == 1 ? "heuristic: solution pool overflow" : "heuristic: backtrace arena overflow"; return WHAMD_ERR_UNSUPPORTED; }
	out.max_solutions = stats[1];
	out.total_solutions = stats[2];
	return WHAMD_OK;
}

}  // namespace whamd
#endif   // WHAMD_DEBUG_BUILD

namespace whamd {

// Allele votes of the final bipartition, optimal phasing and mutations per column (src/pedmecheuristic.cpp:361-406).
void heuristic_finish(const HeurPlan& pl, HeurResult& out) {
	const uint32_t n = pl.n_cols, S = pl.n_samples;
	out.haplotypes.assign((size_t)S * 2 * n, -1);
	out.mutated.assign((size_t)S * 2 * n, 0);
	out.score = 0.0f;
	if (n == 0 || S == 0) return;
	std::vector<float> balances((size_t)n * 2 * S, 0.0f);
	for (uint32_t r = 0; r < pl.n_reads; ++r)
		for (uint64_t v = pl.read_ptr[r]; v < pl.read_ptr[r + 1]; ++v) {
			const int a = pl.var_allele[v];
			if (a >= 0) balances[(size_t)pl.var_col[v] * 2 * S + 2 * pl.read_sample[r] + out.bipartition[r]] += (float)(2 * a - 1) * pl.var_quality[v];
		}
	HeurDev D{};
	D.n_cols = n; D.n_samples = S; D.n_trios = pl.n_trios; D.distrust = pl.distrust;
	D.trios = pl.trios.data();
	D.mutation = pl.mutation.data(); D.genotype = pl.genotype.data();
	for (uint32_t p = 0; p < n; ++p) {
		uint8_t phase[HEUR_MAXS] = {0}, mut[2 * HEUR_MAXS] = {0};
		heur_opt_phasing(D, balances.data() + (size_t)p * 2 * S, out.transmission[p], p, phase, mut);
		for (uint32_t s = 0; s < S; ++s) {
			out.haplotypes[((size_t)s * 2 + 0) * n + p] = (int8_t)(phase[s] & 1);
			out.haplotypes[((size_t)s * 2 + 1) * n + p] = (int8_t)((phase[s] & 2) >> 1);
			out.mutated[((size_t)s * 2 + 0) * n + p] = mut[2 * s];
			out.mutated[((size_t)s * 2 + 1) * n + p] = mut[2 * s + 1];
		}
	}
}

}  // namespace whamd
