// genotype.h -- GenotypeDPTable: the forward-backward genotyper that shares the phasing
// path's columns, indexing scheme and pedigree partitions (src/genotypedptable.cpp:16-451) -- sum-product instead of
// min-plus, no backtrace, no ties.  Device path: genotype_device.hip; f64 arithmetic (the reference computes in long double,
// so parity is to a relative tolerance, not bit-exact).
#pragma once
#include <string>
#include <vector>

#include "problem.h"

namespace whamd {

struct GenotypeStats {
	uint64_t n_columns = 0, n_cells = 0;   // sum_c 2^k_c
	uint64_t launches = 0;
	double backward_ms = 0, forward_ms = 0, total_ms = 0;   // HIP events: checkpoint pass / windows (recompute + forward) / all
	uint32_t window = 0;                   // columns per window (backward columns kept at window ends, recomputed inside)
	uint32_t max_coverage = 0, transmissions = 0;
	uint32_t slot_runs = 0;                // run-fused path: launches per chain (0: per-column kernels)
};

// Per-column model of the genotyper, built on the host from a Problem (columns_only):
//   transition_bern  [n_cols][2 * triples + 1]  normalised Bernoulli terms: P(i -> j) = bern[popcount(i ^ j)]
//                    (TransitionProbabilityComputer, src/transitionprobabilitycomputer.cpp:22-45)
//   allele_prior     [n_cols][T][A]             P(allele assignment a | transmission value i) from the genotype priors
//                    (:48-90: product of the individuals' priors, divided by the multiplicity of the genotype vector, normalised)
//   error_prob       [entries]                  10^(-phred / 10), 0.9999 for phred 0 (src/genotypecolumncostcomputer.cpp:26-48)
struct GenotypeModel {
	uint32_t A = 0;                 // allele assignments = 2^P
	std::vector<double> transition_bern, allele_prior, error_prob;
	std::vector<uint8_t> genotype_index;   // [T][A][n_ind]: allele0 + allele1 of individual under (i, a)  (:376-383)
};
whamd_status_t build_genotype_model(const Problem& p, GenotypeModel& m, std::string& msg);

// gl_out: [n_ind][n_cols][3] genotype likelihoods (0/0, 0/1, 1/1), each triple normalised to sum 1
// (GenotypeDPTable::get_genotype_likelihoods, src/genotypedptable.cpp:444-451).
whamd_status_t genotype_solve_device(const Problem& p, const GenotypeModel& m, int device, uint32_t window_hint,
                                     std::vector<double>& gl_out, GenotypeStats& st, std::string& msg);

// The run-fused path (genotype_slots.hip): slot runs with sums instead of minima, both chains side by side, one combine launch.
// `used` = false (and WHAMD_OK): the table is not eligible, take genotype_solve_device's per-column kernels.
whamd_status_t genotype_solve_slots(const Problem& p, const GenotypeModel& m, int device, std::vector<double>& gl_out, GenotypeStats& st,
                                    bool& used, std::string& msg);

// Frees the device memory genotype_solve_device keeps between calls (one column store per device).
void genotype_release_cache();
// The phasing tables' counterpart (dp_device.hip): one backtrace arena kept between tables; a genotyping call that finds the device
// more than half full gives it back first.
void dptable_release_arena_cache();
// The column store kept between calls (mapping tens of GB of fresh device memory took seconds in one call out of four):
// acquire returns the cached block of `device` grown to `bytes` and marks it in use, or nullptr (in use by another call,
// allocation failed, caching disabled) -- the caller then allocates its own.  release marks it idle again; a block larger than
// a quarter of the device's memory is freed instead of kept (a later phasing solve sizes its arena from what is free).
void* genotype_slab_acquire(int device, size_t bytes);
void genotype_slab_release(int device);
size_t genotype_slab_idle_bytes(int device);

}  // namespace whamd
