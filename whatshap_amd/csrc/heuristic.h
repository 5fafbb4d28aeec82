// heuristic.h -- PedMecHeuristic: the beam-search sibling of PedigreeDPTable behind the same API
// (src/pedmecheuristic.cpp:123-409 solve, :420-622 helpers; whatshap/core.pyx:674-734; selected at whatshap/cli/phase.py:589-603
// for coverages the exact DP cannot afford).  Per column at most row_limit partial solutions (bipartition of the active
// reads, transmission value, float score, per-haplotype allele balances over the window of still-open positions) are
// projected, extended read by read, pruned and scored.
//
// The work per solution is independent (float arithmetic, restated operation by operation so that every decision equals the
// reference's: the scores are IEEE single floats, contraction off); what couples the solutions of a column is order: duplicates
// are merged into their first occurrence, copies are appended in solution order, pruning keeps the survivors in order.  Those
// become hash / scan / radix-select phases.  heuristic_core.h holds the solver ONCE, written against a tiny execution interface
// (thread id, thread count, barrier, atomics): heuristic_device.hip instantiates it as a persistent single-workgroup HIP kernel
// (one launch per table, 1024 threads), heuristic_host.cpp with one thread as the CPU diagnostic the test-suite runs against the
// compiled reference.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/whatshap_amd.h"

namespace whamd {

constexpr uint32_t HEUR_MAX_ROW_LIMIT = 65535;   // MAX_ROW_LIMIT, src/mecheader.h

// Everything the solver reads, flattened on the host (heuristic.cpp: build_heuristic_plan): the per-column bookkeeping of
// solve() that does not depend on the solutions -- which active reads continue, the merged balance vectors of the reads that
// start (src/pedmecheuristic.cpp:199-238), the windows.
struct HeurPlan {
	uint32_t n_cols = 0, n_reads = 0, n_samples = 0, n_trios = 0, tm_bits = 0;
	uint32_t row_limit = 256, distrust = 0;
	uint32_t w_max = 1, act_max = 0, nw = 1;   // widest window, most active reads in a column, words per bipartition
	std::vector<uint32_t> trios;               // [n_trios * 3] sample ranks (as src/pedmecheuristic.cpp:66-70 maps them)
	std::vector<float> recomb, mutation;       // [n_cols] recombCost, mutationCost (:29-37)
	std::vector<int8_t> genotype;              // [n_samples][n_cols] 0 / 1 / 2 (:73-82)
	std::vector<uint32_t> start_index;         // [n_cols + 1] first read starting at column p (:129-137)
	// per column
	std::vector<uint32_t> window, n_kept, kept_off, n_new, new_off;   // new_off: into the per-new-read arrays
	std::vector<uint32_t> kept;                // indices into the previous column's active list
	// per new read (column order)
	std::vector<uint32_t> new_sample;
	std::vector<int32_t> new_equal_to;         // index (within the column's new reads) of the identical read it was merged into, or -1
	std::vector<uint8_t> new_seen, new_useful; // sample seen before this read (:261, :291); "useful" of the trusted-genotype mode (:256-258)
	std::vector<uint64_t> new_bal_off;         // into new_balance: window[p] floats
	std::vector<float> new_balance;
	std::vector<int32_t> new_target;           // parallel to new_balance: the sample's genotype at the window's positions (0 / 1 / 2), as words --
	                                           // the solver reads them as wave-uniform scalars next to the balances (a byte per position
	                                           // out of `genotype` was a dependent memory round trip per position on the device)
	// for the final phasing (host, :361-406)
	std::vector<uint32_t> sample_global_id;    // [n_samples]
	std::vector<uint32_t> positions;           // [n_cols]
	std::vector<uint64_t> read_ptr;            // copies of the view, positions as column indices
	std::vector<uint32_t> var_col;
	std::vector<int8_t> var_allele;
	std::vector<float> var_quality;
	std::vector<uint32_t> read_sample;         // rank per read
};

// What the solver reads per column / per starting read, one 32-byte record each (the device kernel fetches a record with one scalar
// load; as separate arrays they were five dependent loads and ten kernel-argument registers).
struct HeurColMeta { uint32_t window, n_kept, kept_off, n_new, new_off, pad[3]; };
struct HeurReadMeta { uint32_t sample; int32_t equal_to; uint32_t seen, useful; uint64_t bal_off; uint32_t pad[2]; };
static_assert(sizeof(HeurColMeta) == 32 && sizeof(HeurReadMeta) == 32, "one scalar load each");

struct HeurResult {
	float score = 0.0f;                        // getOptScore(): the reference never assigns it, it stays 0 (:19, :338-346)
	std::vector<uint8_t> bipartition;          // [n_reads] getOptBipartition() bits
	std::vector<uint32_t> transmission;        // [n_cols]
	std::vector<int8_t> haplotypes;            // [n_samples][2][n_cols]
	std::vector<uint8_t> mutated;              // [n_samples][2][n_cols]
	uint64_t max_solutions = 0, total_solutions = 0;   // widest column, sum over the columns
	double device_ms = 0.0;
};

std::vector<HeurColMeta> heuristic_col_meta(const struct HeurPlan& plan);
std::vector<HeurReadMeta> heuristic_read_meta(const struct HeurPlan& plan);
whamd_status_t build_heuristic_plan(const whamd_readset_view* rs, const uint32_t* recombcost, size_t n_recombcost, const whamd_pedigree_view* ped,
                                    bool distrust, const uint32_t* positions, size_t n_positions, uint32_t row_limit, bool allow_mutations,
                                    HeurPlan& plan, std::string& msg);
// the beam search: bipartition + transmission (device: heuristic_device.hip; host diagnostic: heuristic_host.cpp)
whamd_status_t heuristic_solve_device(const HeurPlan& plan, int device, HeurResult& out, std::string& msg);
// Several tables of one device in flight (heuristic_device.hip): enqueue() uploads the plans and submits ONE launch whose grid is the
// tables (one persistent workgroup each) on the batch's own stream and returns; wait() collects every table (and runs a table again whose
// records outgrew their arena).  The plans must outlive the batch.
class HeurBatch {
public:
	HeurBatch();
	~HeurBatch();
	HeurBatch(const HeurBatch&) = delete;
	HeurBatch& operator=(const HeurBatch&) = delete;
	whamd_status_t enqueue(const HeurPlan* const* plans, size_t n, int device, std::string& msg);
	whamd_status_t wait(HeurResult* outs, std::string& msg);   // outs[n]
	struct Impl;
private:
	Impl* impl_;
};
// gives the device buffers kept between solves back to the driver (whamd_release_caches)
void heuristic_release_cache();
whamd_status_t heuristic_solve_host(const HeurPlan& plan, HeurResult& out, std::string& msg);
// allele votes of the final bipartition and the optimal phasing per column (host, src/pedmecheuristic.cpp:361-406)
void heuristic_finish(const HeurPlan& plan, HeurResult& out);

}  // namespace whamd
