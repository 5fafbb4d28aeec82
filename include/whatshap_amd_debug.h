/*
 * whatshap_amd_debug.h -- TEST INFRASTRUCTURE, not part of the drop-in boundary.
 *
 * Entry points that exist only in libwhatshap_amd_debug.so (the product sources compiled with -DWHAMD_DEBUG_BUILD, `make debug`):
 * CPU emulators of the run plans and the single-thread host instantiation of the heuristic, which the CPU test-suite compares with
 * the oracle.  The debug library also carries the kernel instantiations with in-kernel cycle stamps and the timing switches
 * (WHAMD_SLOT_STAMPS, WHAMD_SLOT_SKIP: results invalid) that scripts/gpu_slot_*.py use.  libwhatshap_amd.so has none of this.
 */
#ifndef WHATSHAP_AMD_DEBUG_H
#define WHATSHAP_AMD_DEBUG_H

#include "whatshap_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host-only diagnostic of the slot-run planner (no device needed, small inputs only): builds the forward plan of a
 * single-individual table exactly as whamd_dptable_create would (slot_l local slots preferred -- add 100 for 8 instead of 4 cells per thread --, symmetry level) and
 * executes it cell by cell on the CPU the way the kernels do -- same physical cell indices, decision bits, record
 * layout, exchange layouts and mirror rules.  index_out[n_columns]: the index path (index_path[c].index,
 * src/pedigreedptable.h:17-21), score_out: the optimal score.  Lets the CPU test-suite check the PLAN against the
 * oracle; not a solver and never used by one (WHAMD_ERR_UNSUPPORTED for pedigrees). */
whamd_status_t whamd_debug_emulate_slot_plan(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                            const whamd_pedigree_view* pedigree, int distrust_genotypes,
                                            const uint32_t* positions, size_t n_positions, int slot_l, int symmetry,
                                            uint32_t* index_out, uint32_t* score_out, uint64_t* n_run_columns_out);

/* The same diagnostic for a pedigree table with one or two trios (T = 4 / 16): the pedigree slot plan (one (cell,
 * transmission value) per lane, cost forms split into per-workgroup / per-wave / per-lane tables, butterfly min-plus
 * step, one record byte per lane and column) executed on the CPU as kernels_pedslots.h does it.  slot_l <= 0: the
 * default number of local slots, else that many.  transmission_out[n_columns]: index_path[c].inheritance_value.
 * WHAMD_ERR_UNSUPPORTED when the table is not eligible for pedigree slot runs. */
whamd_status_t whamd_debug_emulate_pedslot_plan(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                               const whamd_pedigree_view* pedigree, int distrust_genotypes,
                                               const uint32_t* positions, size_t n_positions, int slot_l,
                                               uint32_t* index_out, uint32_t* transmission_out, uint32_t* score_out,
                                               uint64_t* n_run_columns_out);

/* Host-only check of the LAZY generic term lists (csrc/problem.cpp build_problem `lazy_fact_terms` + fill_lazy_terms: what whamd_dptable_create does for a
 * trio / quartet with untrusted genotypes whose runs read the factorised line): the problem is built twice -- every term list at once, and lazily with the
 * lists of the columns whose bit is set in need[n_columns] (NULL: every column) filled in afterwards, in `rounds` calls (the columns dealt out round robin) --
 * and the two are compared term by term on the needed columns.  *lazy_out: 1 if the table took the lazy route at all (0: not a factorised table, nothing to
 * compare); *differences_out: columns whose lists differ (0 expected); *built_before_out / *built_after_out: columns with term lists before / after the fills. */
whamd_status_t whamd_debug_lazy_terms_check(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                            const whamd_pedigree_view* pedigree, int distrust_genotypes,
                                            const uint32_t* positions, size_t n_positions, const uint8_t* need, int rounds,
                                            int* lazy_out, uint64_t* differences_out, uint64_t* built_before_out, uint64_t* built_after_out);

/* HOST-ONLY DIAGNOSTIC: the same solver source run with one CPU thread (csrc/heuristic_host.cpp), for the CPU test-suite to
 * compare with the compiled reference; never what the drop-in class calls. */
whamd_status_t whamd_debug_pedmec_heuristic_create_host(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                                        const whamd_pedigree_view* pedigree, int distrust_genotypes,
                                                        const uint32_t* positions, size_t n_positions, uint32_t row_limit, int allow_mutations,
                                                        whamd_heuristic** out);

#ifdef __cplusplus
}
#endif

#endif /* WHATSHAP_AMD_DEBUG_H */
