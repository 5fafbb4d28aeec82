// problem.h -- host-side model of one PedigreeDPTable instance: the flattened ReadSet (columns in
// CSR form), the per-column indexing scheme, the pedigree partitions and the per-column cost terms.
// Everything here is precomputed once on the host and uploaded; the device never sees Read/Entry objects.
//
// Reference counterparts (whatshap/whatshap @ 2025-07-11):
//   ColumnIterator            src/columniterator.cpp:10-59,91-169   -> build_columns()
//   ColumnIndexingScheme      src/columnindexingscheme.cpp:7-34,62-85 -> k/b/f/forward mask per column
//   PedigreePartitions        src/pedigreepartitions.cpp:7-42        -> h2p
//   PedigreeColumnCostComputer ctor  src/pedigreecolumncostcomputer.cpp:14-50 -> cost terms
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/whatshap_amd.h"
#include "host_parallel.h"

namespace whamd {

constexpr uint32_t INF = 0xFFFFFFFFu;
constexpr int MAX_COVERAGE = 25;   // device limit on reads per column (the CLI caps at 23, cli/phase.py:1181)
constexpr int MAX_IND = 6;         // individuals per pedigree of the templated kernels (and of the genotyping path)
constexpr int MAX_T = 16;          // transmission values = 4^triples of the templated kernels (<= 2 trios)
constexpr int MAX_IND_WIDE = 12;   // phasing: larger pedigrees run on the generic per-column kernel (column_step_wide)
constexpr int MAX_TRIPLES_WIDE = 3;
constexpr int MAX_T_WIDE = 64;     // 4^3

struct ColumnEntry {  // Entry (src/entry.h) plus the individual index of its read
	uint32_t read_id;
	uint32_t phred;
	uint8_t allele;  // 0 REF, 1 ALT, 2 BLANK
	uint8_t sample;  // individual index (read_sources, src/pedigreedptable.cpp:32-34)
};

// One candidate of min_a in get_cost() (src/pedigreecolumncostcomputer.cpp:101-114), rewritten
// per individual: value(x) = c + sum_s sigma_s * L_s(x), sigma in {-1, 0, +1}, where
// L_s(x) = sum over the set bits of x that belong to reads of individual s of d_bit, d = +q (REF),
// -q (ALT), 0 (BLANK)  (DESIGN.md "cost closed form").
struct CostTerm {
	uint32_t c;      // constant part (u32, modular)
	uint32_t plus;   // bit s set: + L_s(x)
	uint32_t minus;  // bit s set: - L_s(x)
};

struct Problem {
	// ---- inputs (copied from the views)
	uint32_t n_reads = 0;
	std::vector<uint64_t> read_ptr;
	RawVec<int32_t> var_position;   // (RawVec: sized without being zero-filled, host_parallel.h.  Since round 6 the three var_* arrays stay EMPTY: build_problem reads the view's)
	RawVec<uint8_t> var_allele;
	RawVec<uint32_t> var_quality;
	std::vector<uint32_t> read_source;
	std::vector<uint32_t> read_first_col, read_last_col;   // [n_reads] columns of a read's first and last variant
	uint32_t n_ind = 0, n_triples = 0, n_variants = 0;
	std::vector<uint32_t> individual_id;
	std::vector<std::array<uint32_t, 3>> triples;  // by individual index
	std::vector<uint8_t> genotype;
	std::vector<double> gl;
	bool have_gl = false;
	bool distrust = false;
	std::vector<uint32_t> recomb;     // padded to n_cols
	std::vector<uint32_t> positions;  // n_cols
	// ---- derived
	uint32_t n_cols = 0;
	uint32_t T = 1, P = 0;
	std::vector<int8_t> h2p;  // [T][n_ind][2]
	std::vector<uint64_t> col_ptr;  // [n_cols + 1]
	RawVec<ColumnEntry> entries;
	std::vector<uint8_t> k, b, f;        // per column
	std::vector<uint32_t> fwd_mask;      // per column: bits of reads that continue into the next column
	// cost terms per (column, transmission value)
	std::vector<uint64_t> term_ptr;  // [n_cols * T + 1] offsets into terms
	RawVec<CostTerm> terms;
	// A trio whose genotypes are not trusted: the minimum over the 16 allele assignments of a transmission value factorises over the
	// individuals (build_problem; kernels_pedslots.h PSLOT_FACT) -- 16 entries per (column, transmission value): three signed sums
	// {L_X, L_Y, L_child, 0} and twelve constants.  Empty when the table is not of that shape.
	RawVec<CostTerm> fterms;         // [n_cols * T * 16]
	// fterm_kind 2: a QUARTET (two children of the same two founders) whose genotypes are not trusted -- in haplotype space the line does not depend on the
	// transmission value: fterms is [n_cols * 20] (four signed sums L_X, L_Y, L_C1, L_C2, then the costs of X, Y, C1, C2 carrying (h0, h1) = 00, 01, 10, 11), and the
	// value only wires the children to the founders' haplotypes: fact4_roles = {u_1 | v_1 << 16, (u_1 == u_2) | (v_1 == v_2) << 16}, bit t of each 16-bit mask
	// (slots.h PSLOT_FACT4).
	uint32_t fterm_kind = 0;         // 0 none, 1 trio (16 entries per (column, value)), 2 quartet (20 per column)
	bool lazy_terms = false;         // generic terms exist only for the columns of `terms_built` (build_problem's sample, fill_lazy_terms' additions)
	std::vector<uint8_t> terms_built;
	uint32_t fact4_roles[2] = {0, 0};
	// per column and individual: signed per-bit deltas d (REF +q, ALT -q, BLANK 0) of L_s(x) - R_s
	RawVec<int32_t> delta;  // [col_ptr[c] * n_ind ... ): for column c, delta[(col_ptr[c] * n_ind) + s * k_c + bit]
	uint32_t max_k = 0;
	uint64_t n_cells = 0, algorithmic_bytes = 0;
	double value_bound = 0.0;  // upper bound on every finite DP value

	const ColumnEntry* col_begin(uint32_t c) const { return entries.data() + col_ptr[c]; }
	uint64_t term_begin(uint32_t c, uint32_t t) const { return term_ptr[(size_t)c * T + t]; }
	uint64_t term_end(uint32_t c, uint32_t t) const { return term_ptr[(size_t)c * T + t + 1]; }
};

// Builds the problem; on failure returns a status != WHAMD_OK and sets msg.
whamd_status_t build_problem(const whamd_readset_view* rs, const uint32_t* recombcost, size_t n_recombcost,
                             const whamd_pedigree_view* ped, bool distrust, const uint32_t* positions,
                             size_t n_positions, Problem& out, std::string& msg, bool columns_only = false, bool lazy_fact_terms = false);
// `lazy_fact_terms`: a table whose runs will read the factorised line (Problem::fterms: a trio or a quartet with untrusted genotypes) gets its generic term
// lists only for a sample of columns (on which the line is checked); fill_lazy_terms(p, need) builds them for the columns that turn out to need them -- those
// the planner leaves outside runs, or all of them when the table takes another path.  150 MB less to build, join and upload for a quartet of 50 000 columns.
whamd_status_t fill_lazy_terms(Problem& p, const std::vector<uint8_t>& need, std::string& msg);

// Host part of get_super_reads (src/pedigreedptable.cpp:344-388 + get_alleles,
// src/pedigreecolumncostcomputer.cpp:117-175) and get_optimal_partitioning (:391-406) from a finished path.
struct Solution {
	uint32_t optimal_score = 0;
	std::vector<uint32_t> path_index, path_trans;  // n_cols
	std::vector<uint8_t> allele0, allele1;         // [n_ind * n_cols]
	std::vector<uint32_t> quality;                 // [n_ind * n_cols]
	std::vector<uint8_t> partition;                // [n_reads]
	bool superreads_done = false;                  // allele0 / allele1 / quality came from the device (dp_device.hip: a single individual, trusted genotypes): finish_solution only makes the partition
};
whamd_status_t finish_solution(const Problem& p, Solution& s, std::string& msg);

}  // namespace whamd
