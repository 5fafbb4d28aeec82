// heuristic.cpp -- host side of the PedMecHeuristic drop-in (heuristic.h): everything of the constructor and of solve() that does
// not depend on the beam -- positions, float cost vectors, sample ranks, genotypes, the active-read bookkeeping and the merged
// balance vectors of every column -- flattened once for the solver (src/pedmecheuristic.cpp:9-83, :123-238).
#include "heuristic.h"

#pragma clang fp contract(off)

#include <algorithm>
#include <cmath>
#include <limits>
#include <map>
#include <set>
#include <unordered_map>

namespace whamd {

whamd_status_t build_heuristic_plan(const whamd_readset_view* rs, const uint32_t* recombcost, size_t n_recombcost, const whamd_pedigree_view* ped,
                                    bool distrust, const uint32_t* positions, size_t n_positions, uint32_t row_limit, bool allow_mutations,
                                    HeurPlan& pl, std::string& msg) {
	pl = HeurPlan();
	if (!rs || !ped) { msg = "null argument"; return WHAMD_ERR_INVALID; }
	pl.row_limit = std::min<uint32_t>(row_limit, HEUR_MAX_ROW_LIMIT);   // :15
	if (pl.row_limit == 0) { msg = "row_limit must be positive"; return WHAMD_ERR_INVALID; }
	pl.distrust = distrust ? 1u : 0u;
	const uint32_t m = rs->n_reads;
	pl.n_reads = m;
	const uint64_t nnz = m ? rs->read_ptr[m] : 0;
	// positions (:40-47; ReadSet::get_positions, src/readset.cpp:54-62)
	if (positions) pl.positions.assign(positions, positions + n_positions);
	else {
		std::set<uint32_t> all;
		for (uint64_t i = 0; i < nnz; ++i) all.insert((uint32_t)rs->var_position[i]);
		pl.positions.assign(all.begin(), all.end());
	}
	const uint32_t n = (uint32_t)pl.positions.size();
	pl.n_cols = n;
	std::unordered_map<uint32_t, uint32_t> pos_map;
	for (uint32_t i = 0; i < n; ++i) pos_map[pl.positions[i]] = i;
	if (n_recombcost != n) {   // (the reference indexes recombCost[p] for every column and mutationCost[size - 1])
		msg = "PedMecHeuristic: recombcost must have one entry per position (" + std::to_string(n_recombcost) + " given, " + std::to_string(n) + " positions)";
		return WHAMD_ERR_INVALID;
	}
	// costs (:28-38): recombCost[0] stays 0; the mutation costs are formed in double and rounded to float
	pl.recomb.assign(n, 0.0f);
	pl.mutation.assign(n, std::numeric_limits<float>::infinity());
	for (uint32_t i = 1; i < n; ++i) {
		pl.recomb[i] = (float)recombcost[i];
		if (allow_mutations) pl.mutation[i - 1] = (float)(0.75 * (pl.recomb[i - 1] + pl.recomb[i]));
	}
	if (allow_mutations && n) pl.mutation[n - 1] = (float)(pl.recomb[n - 1] * 1.5);
	// reads: every position must be a column; reads sorted by their first position (the CLI sorts the ReadSet first, cli/phase.py:590)
	pl.read_ptr.assign(rs->read_ptr, rs->read_ptr + m + 1);
	pl.var_col.resize(nnz);
	pl.var_allele.resize(nnz);
	pl.var_quality.resize(nnz);
	for (uint64_t i = 0; i < nnz; ++i) {
		const auto it = pos_map.find((uint32_t)rs->var_position[i]);
		if (it == pos_map.end()) { msg = "PedMecHeuristic: a read covers position " + std::to_string(rs->var_position[i]) + " which is not among the positions to phase"; return WHAMD_ERR_INVALID; }
		pl.var_col[i] = it->second;
		pl.var_allele[i] = (int8_t)rs->var_allele[i];
		pl.var_quality[i] = (float)rs->var_quality[i];
	}
	std::vector<uint32_t> first_col(m), last_col(m);
	for (uint32_t r = 0; r < m; ++r) {
		if (rs->read_ptr[r + 1] <= rs->read_ptr[r]) { msg = "PedMecHeuristic: empty read"; return WHAMD_ERR_INVALID; }
		first_col[r] = pl.var_col[rs->read_ptr[r]];
		last_col[r] = pl.var_col[rs->read_ptr[r + 1] - 1];
		if (r && first_col[r] < first_col[r - 1]) { msg = "PedMecHeuristic: reads in ReadSet are not sorted."; return WHAMD_ERR_INVALID; }
	}
	// samples (:49-71): ranks of the ids of the reads and of the trio members.  The reference takes the trios as INDIVIDUAL INDICES
	// (Pedigree::get_triples) and looks genotypes up by rank: ids, indices and ranks must coincide, as they do for pedigrees built in id order.
	std::map<uint32_t, uint32_t> id_to_index;
	for (uint32_t i = 0; i < ped->n_individuals; ++i) id_to_index[ped->individual_id[i]] = i;   // (later insertions win, src/pedigree.cpp:47-55)
	std::set<uint32_t> sample_set;
	for (uint32_t r = 0; r < m; ++r) sample_set.insert((uint32_t)rs->read_sample_id[r]);
	std::vector<uint32_t> trio_index;
	for (uint32_t t = 0; t < ped->n_triples; ++t)
		for (int q = 0; q < 3; ++q) {
			const auto it = id_to_index.find(ped->triple_ids[3 * t + q]);
			if (it == id_to_index.end()) { msg = "PedMecHeuristic: a trio names an individual the pedigree does not hold"; return WHAMD_ERR_INVALID; }
			trio_index.push_back(it->second);
			sample_set.insert(it->second);
		}
	pl.sample_global_id.assign(sample_set.begin(), sample_set.end());
	pl.n_samples = (uint32_t)pl.sample_global_id.size();
	pl.n_trios = ped->n_triples;
	pl.tm_bits = 2 * pl.n_trios;
	if (pl.n_samples == 0 || pl.n_samples > 8 || pl.n_trios > 8 || pl.tm_bits > 8) {
		msg = "PedMecHeuristic on the device: at most 8 samples and 4 trios (" + std::to_string(pl.n_samples) + " samples, " + std::to_string(pl.n_trios) + " trios)";
		return n ? WHAMD_ERR_UNSUPPORTED : WHAMD_OK;
	}
	std::unordered_map<uint32_t, uint32_t> sample_map;
	for (uint32_t i = 0; i < pl.n_samples; ++i) sample_map[pl.sample_global_id[i]] = i;
	for (uint32_t v : trio_index) pl.trios.push_back(sample_map[v]);
	if (pl.n_samples > ped->n_individuals || n > ped->n_variants) { msg = "PedMecHeuristic: the pedigree holds fewer individuals / variants than the reads and positions need"; return WHAMD_ERR_INVALID; }
	pl.genotype.assign((size_t)pl.n_samples * n, 0);
	for (uint32_t s = 0; s < pl.n_samples; ++s)
		for (uint32_t p = 0; p < n; ++p) {
			const uint8_t g = ped->genotype[(size_t)s * ped->n_variants + p];   // get_genotype(rank, column), :78
			if (g > 2) { msg = "PedMecHeuristic: every genotype must be 0/0, 0/1 or 1/1"; return WHAMD_ERR_INVALID; }
			pl.genotype[(size_t)s * n + p] = (int8_t)g;
		}
	pl.read_sample.resize(m);
	for (uint32_t r = 0; r < m; ++r) pl.read_sample[r] = sample_map[(uint32_t)rs->read_sample_id[r]];
	// first read starting at column p (:129-137)
	pl.start_index.assign(1, 0);
	{
		uint32_t q = 0;
		for (uint32_t p = 0; p < n; ++p) {
			while (q < m && first_col[q] <= p) ++q;
			pl.start_index.push_back(q);
		}
	}
	std::vector<uint8_t> seen(pl.n_samples, 0);   // (:140-142) children count as seen
	for (uint32_t t = 0; t < pl.n_trios; ++t) seen[pl.trios[3 * t + 2]] = 1;
	// the columns (:154-238)
	std::vector<uint32_t> active;
	uint32_t right = 0;
	pl.window.resize(n); pl.n_kept.resize(n); pl.kept_off.resize(n); pl.n_new.resize(n); pl.new_off.resize(n);
	for (uint32_t p = 0; p < n; ++p) {
		std::vector<uint32_t> next;
		pl.kept_off[p] = (uint32_t)pl.kept.size();
		for (uint32_t i = 0; i < active.size(); ++i)
			if (last_col[active[i]] >= p) { next.push_back(active[i]); pl.kept.push_back(i); }
		pl.n_kept[p] = (uint32_t)next.size();
		right = std::max(right, p);
		for (uint32_t r = pl.start_index[p]; r < pl.start_index[p + 1]; ++r) right = std::max(right, last_col[r]);
		const uint32_t w = right + 1 - p;
		pl.window[p] = w;
		pl.w_max = std::max(pl.w_max, w);
		const uint32_t num_new = pl.start_index[p + 1] - pl.start_index[p];
		pl.n_new[p] = num_new;
		pl.new_off[p] = (uint32_t)pl.new_sample.size();
		std::vector<std::vector<float>> balances;
		std::vector<int32_t> equal_to(num_new, -1);
		std::vector<uint32_t> sample_ids;
		for (uint32_t i = 0; i < num_new; ++i) {
			const uint32_t r = pl.start_index[p] + i;
			next.push_back(r);
			std::vector<float> b(w, 0.0f);
			sample_ids.push_back(pl.read_sample[r]);
			for (uint64_t v = rs->read_ptr[r]; v < rs->read_ptr[r + 1]; ++v) {
				const uint32_t o = pl.var_col[v] - p;
				const int a = pl.var_allele[v];
				const float q = pl.var_quality[v];
				b[o] += q * (float)a - q * (float)(1 - a);   // :211
			}
			for (uint32_t j = 0; j < i; ++j) {   // identical reads of one sample are summarised (:213-229)
				if (equal_to[j] != -1 || sample_ids[j] != sample_ids[i]) continue;
				bool equal = true;
				for (uint32_t k = 0; k < w; ++k)
					if (balances[j][k] * b[k] < 0 || (balances[j][k] != 0.0f) != (b[k] != 0.0f)) { equal = false; break; }
				if (equal) {
					equal_to[i] = (int32_t)j;
					for (uint32_t k = 0; k < w; ++k) balances[j][k] += b[k];
					break;
				}
			}
			balances.push_back(b);
		}
		for (uint32_t i = 0; i < num_new; ++i) {
			const uint32_t s = sample_ids[i];
			pl.new_sample.push_back(s);
			pl.new_equal_to.push_back(equal_to[i]);
			pl.new_seen.push_back(seen[s]);
			bool useful = false;   // trusted genotypes (:256-258): a heterozygous position the read says something about
			for (uint32_t j = 0; j < w && !useful; ++j) useful = pl.genotype[(size_t)s * n + p + j] == 1 && balances[i][j] != 0;
			pl.new_useful.push_back(useful ? 1 : 0);
			pl.new_bal_off.push_back(pl.new_balance.size());
			pl.new_balance.insert(pl.new_balance.end(), balances[i].begin(), balances[i].end());
			for (uint32_t j = 0; j < w; ++j) pl.new_target.push_back((int32_t)pl.genotype[(size_t)s * n + p + j]);
			seen[s] = 1;
		}
		active.swap(next);
		pl.act_max = std::max<uint32_t>(pl.act_max, (uint32_t)active.size());
	}
	pl.nw = std::max(1u, (pl.act_max + 31u) / 32u);
	return WHAMD_OK;
}

std::vector<HeurColMeta> heuristic_col_meta(const HeurPlan& pl) {
	std::vector<HeurColMeta> out(pl.n_cols);
	for (uint32_t p = 0; p < pl.n_cols; ++p) out[p] = HeurColMeta{pl.window[p], pl.n_kept[p], pl.kept_off[p], pl.n_new[p], pl.new_off[p], {0, 0, 0}};
	return out;
}

std::vector<HeurReadMeta> heuristic_read_meta(const HeurPlan& pl) {
	std::vector<HeurReadMeta> out(pl.new_sample.size());
	for (size_t r = 0; r < out.size(); ++r) out[r] = HeurReadMeta{pl.new_sample[r], pl.new_equal_to[r], pl.new_seen[r], pl.new_useful[r], pl.new_bal_off[r], {0, 0}};
	return out;
}

}  // namespace whamd
