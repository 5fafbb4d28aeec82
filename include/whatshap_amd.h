/*
 * whatshap_amd.h -- C ABI of the MI355X-native wMEC / PedMEC solver.
 *
 * This is the drop-in boundary for the hot path of `whatshap phase`: the C++ class
 * PedigreeDPTable that whatshap/core.pyx:364-416 binds through whatshap/cpp.pxd:85-90
 *
 *     PedigreeDPTable(ReadSet*, vector[unsigned int] recombcost, Pedigree*, bool distrust_genotypes,
 *                     vector[unsigned int]* positions) except +
 *     void get_super_reads(vector[ReadSet*]*, vector[unsigned int]* transmission_vector) except +
 *     int  get_optimal_score() except +
 *     vector[bool]* get_optimal_partitioning()
 *
 * Every entry point below replaces one of those four members (cited per function).  The ABI uses
 * plain pointers and sizes only: a ReadSet is handed over as a CSR "view" of what
 * src/readset.h / src/read.h store, a Pedigree as a view of what src/pedigree.h stores.  No torch,
 * no C++ types, no ownership transfer: every input pointer is borrowed for the duration of the call
 * that takes it (the table copies what it needs, unlike the reference, which keeps ReadSet* /
 * Pedigree* and re-reads them in get_super_reads, src/pedigreedptable.h:80-86).
 *
 * Error model: functions that can fail return a whamd_status_t; the message a Python binding
 * should raise as RuntimeError (the reference's `except +` path) is returned by
 * whamd_last_error().  The two messages of the reference's hot path are reproduced verbatim:
 *   "Error: Mendelian conflict"                          (src/pedigreedptable.cpp:302)
 *   "ColumnIterator: reads in ReadSet are not sorted."   (src/columniterator.cpp:29)
 *
 * Device selection: one table lives on one HIP device (one process per GPU, or one worker
 * thread per device); independent tables never communicate (no RCCL on this path).
 */
#ifndef WHATSHAP_AMD_H
#define WHATSHAP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: whamd_plan_summary grew (n_yform_runs, n_fact_runs), whamd_solve_stats.group_tables / host_flatten_ms: a caller built against
 * version 1 must not be handed the larger structs -- bindings compare whamd_abi_version() with the header they were built from. */
#define WHAMD_ABI_VERSION 2

/* allele codes, identical to Entry::allele_t (src/entry.h:8) */
#define WHAMD_ALLELE_REF 0
#define WHAMD_ALLELE_ALT 1
#define WHAMD_ALLELE_BLANK 2
#define WHAMD_ALLELE_EQUAL_SCORES 3

/* genotype codes of the pedigree view: canonical index of a diploid bi-allelic genotype
 * (src/genotype.h:22-27: 0 -> 0/0, 1 -> 0/1, 2 -> 1/1); anything else (other ploidy, other
 * alleles, empty genotype) is WHAMD_GT_OTHER and is compatible with no allele assignment,
 * exactly as Genotype::operator!= on the 64-bit code behaves (src/genotype.cpp:148-150). */
#define WHAMD_GT_OTHER 255

typedef enum whamd_status_t {
	WHAMD_OK = 0,
	WHAMD_ERR_INVALID = 1,           /* malformed input (message says what) */
	WHAMD_ERR_MENDELIAN_CONFLICT = 2, /* "Error: Mendelian conflict" */
	WHAMD_ERR_UNSORTED = 3,          /* ColumnIterator sortedness errors */
	WHAMD_ERR_UNSUPPORTED = 4,       /* outside the limits of the device path (coverage > 23, ...) */
	WHAMD_ERR_DEVICE = 5,            /* HIP runtime error / no device / extension missing */
	WHAMD_ERR_OVERFLOW = 6,          /* costs could exceed 32 bits; reference behaviour undefined there */
	WHAMD_ERR_HOST = 7               /* a host-side failure inside the library: out of memory, no thread could be started (message says what) */
} whamd_status_t;

/* View of a ReadSet (src/readset.h:14-26, src/read.h:10-83).  Reads in ReadSet order; the
 * variants of read r are entries read_ptr[r] .. read_ptr[r+1]-1 of the three var_* arrays. */
typedef struct whamd_readset_view {
	uint32_t n_reads;
	const uint64_t* read_ptr;       /* [n_reads + 1] */
	const int32_t* var_position;    /* Read::getPosition   */
	const uint8_t* var_allele;      /* Read::getAllele (0 = REF, 1 = ALT) */
	const uint32_t* var_quality;    /* Read::getVariantQuality (phred) */
	const int32_t* read_sample_id;  /* [n_reads] Read::getSampleID (numeric sample id) */
} whamd_readset_view;

/* View of a Pedigree (src/pedigree.h:16-86).  Individuals in insertion order (that order is
 * the "individual index"); triples by numeric id as passed to Pedigree::addRelationship. */
typedef struct whamd_pedigree_view {
	uint32_t n_individuals;
	const uint32_t* individual_id; /* [n_individuals] */
	uint32_t n_triples;
	const uint32_t* triple_ids;    /* [3 * n_triples]: father id, mother id, child id */
	uint32_t n_variants;           /* Pedigree::get_variant_count() (0 if no individual) */
	const uint8_t* genotype;       /* [n_individuals * n_variants] WHAMD genotype codes */
	const double* genotype_likelihoods; /* [n_individuals * n_variants * 3] phred GL of 0/0, 0/1, 1/1, or NULL */
	const uint8_t* gl_present;     /* [n_individuals * n_variants] 1 if the GL triple is set, or NULL (= all set iff genotype_likelihoods != NULL) */
} whamd_pedigree_view;

typedef struct whamd_dptable whamd_dptable; /* opaque */

/* Per-solve measurements taken with HIP events on the table's own stream. */
typedef struct whamd_solve_stats {
	uint64_t n_columns;
	uint64_t n_cells;            /* sum_c 2^k_c (unique bipartitions) */
	uint64_t n_costs;            /* n_cells * T */
	uint64_t algorithmic_bytes;  /* sum_c 4*T*2^b_c + 12*T*2^f_c + 12*k_c   */
	uint64_t forward_launches;   /* launches of the dominant (column-step) kernel */
	double forward_ms;           /* HIP-event time of the forward pass (all column steps) */
	double backtrace_ms;         /* HIP-event time of the device backtrace */
	double total_ms;             /* HIP-event time forward + backtrace + path download */
	double host_prepare_ms;      /* wall: flattening + descriptor build + upload (outside total_ms) */
	double host_finish_ms;       /* wall: superreads + partitioning on the host */
	uint32_t max_coverage;       /* max_c k_c */
	uint32_t transmissions;      /* T = 4^triples */
	uint32_t bt_chunks;          /* chunked speculative backtrace: chunks walked at once (0: the sequential walk was used) */
	uint32_t bt_missed;          /*   chunks whose true entry state was none of the guesses */
	uint32_t bt_rewalked;        /*   units walked again from the true state */
	uint32_t group_tables;       /* tables that shared this solve's forward launches (whamd_dptable_enqueue_many groups tables of one device;
	                              * 1: the table ran alone).  forward_ms / forward_launches then describe the GROUP's launches. */
	double host_flatten_ms;      /* the part of host_prepare_ms spent flattening the ReadSet (ColumnIterator's work, src/columniterator.cpp:91-169):
	                              * host_prepare_ms - host_flatten_ms = indexing scheme, cost terms, plan and upload -- what the reference's constructor
	                              * does before compute_table (src/pedigreedptable.cpp:15-37; SURVEY.md 8d puts it inside the solve time) */
} whamd_solve_stats;

/* Library / device introspection. */
int whamd_abi_version(void);
/* number of visible HIP devices (0 if none; never fails) */
int whamd_device_count(void);
/* PCI bus id of HIP device `device` ("0000:c5:00.0") into `out` (at most `capacity` bytes incl. the terminator): what a launcher needs to find
 * the NUMA node / CPU list of the device in sysfs (/sys/bus/pci/devices/<id>/numa_node, local_cpulist) and keep rank r's host threads next
 * to GPU r (whatshap_amd.blocks.bind_rank_to_device_cpus).  No reference counterpart: the reference has no device.  WHAMD_ERR_DEVICE if
 * there is no such device, WHAMD_ERR_INVALID for a null / too small buffer. */
whamd_status_t whamd_device_pci_bus_id(int device, char* out, size_t capacity);
/* thread-local message of the last failing call on this thread ("" if none) */
const char* whamd_last_error(void);

/* Frees everything the table holds on the device (buffers, stream, events) while keeping the solution: the getters
 * below stay valid, a later enqueue/solve uploads again.  For callers that keep thousands of solved blocks alive
 * (one table per connected component of a chromosome).  No reference counterpart: the reference frees its
 * backtrace tables only in ~PedigreeDPTable (src/pedigreedptable.cpp:40-48). */
whamd_status_t whamd_dptable_release_device(whamd_dptable* table);

/* whamd_dptable_enqueue for several tables at once: their launch sequences are submitted round robin, a few launches
 * per table and turn, so that the streams of independent blocks fill up side by side and the blocks overlap on the
 * device from the first column on (submitting table after table lets each one run alone for as long as the host needs
 * to submit the next).  Collect every table with whamd_dptable_wait.  No reference counterpart (the reference solves
 * blocks one after the other, cli/phase.py:604). */
whamd_status_t whamd_dptable_enqueue_many(whamd_dptable* const* tables, size_t n_tables);
/* whamd_dptable_wait for several tables: every table's device side is collected, then the host side of all of them (superreads and
 * partitioning: what get_super_reads / get_optimal_partitioning compute, src/pedigreedptable.cpp:344-406) runs on a few host threads at
 * once.  Returns the first failure; every table has left the "in flight" state afterwards. */
whamd_status_t whamd_dptable_wait_many(whamd_dptable* const* tables, size_t n_tables);

/*
 * Replaces PedigreeDPTable::PedigreeDPTable (src/pedigreedptable.cpp:15-37), split in two so that
 * a benchmark can time the device part alone:
 *
 *   whamd_dptable_create : ColumnIterator construction + validation (src/columniterator.cpp:10-59),
 *                          ColumnIndexingScheme per column (src/columnindexingscheme.cpp:7-34,62-85),
 *                          PedigreePartitions (src/pedigreepartitions.cpp:7-42), allele-assignment
 *                          tables (src/pedigreecolumncostcomputer.cpp:14-50); uploads them to `device`.
 *   whamd_dptable_solve  : compute_table() (src/pedigreedptable.cpp:84-174) on the device:
 *                          forward pass over all columns, backtrace -> index path; then the host
 *                          part of get_super_reads / get_optimal_partitioning is evaluated once and
 *                          cached.  May be called repeatedly (re-solves from the uploaded input).
 *
 * recombcost has n_recombcost entries; the reference indexes recombcost[column] without a bounds
 * check (src/pedigreedptable.cpp:289) -- here a missing tail is padded with the last given value
 * (0 if empty).  positions == NULL means ReadSet::get_positions() (src/readset.cpp:54-62).
 * Does NOT mutate the caller's ReadSet (the reference's reassignReadIds(), :24, has no
 * counterpart on a view).
 */
whamd_status_t whamd_dptable_create(const whamd_readset_view* readset, const uint32_t* recombcost,
                                    size_t n_recombcost, const whamd_pedigree_view* pedigree,
                                    int distrust_genotypes, const uint32_t* positions, size_t n_positions,
                                    int device, whamd_dptable** out);
/* whamd_dptable_create -- the replacement of PedigreeDPTable::PedigreeDPTable (src/pedigreedptable.cpp:15-37, see above) -- with solver
 * options (the keys of whamd_dptable_set_option) applied BEFORE the plan is made and uploaded: one upload instead of two.  What it is for: tables that will share their launches with many others (whamd_dptable_enqueue_many) do
 * better with eight cells per thread and twelve local slots when they are wide: "shared_launches" = "1" says so (the library applies the
 * layout to single-individual tables of coverage >= 18: half the wavefronts per table, 24 coverage-20 tables 7.7 M columns/s instead of
 * 6.4 M), while a table solved alone is faster with the default four cells (2.26 M against 1.87 M) and narrow tables gain nothing.
 * "host_threads" = "n" (this call only): how many host threads the create may use -- a caller that creates many tables on threads of its
 * own keeps each to a few. */
whamd_status_t whamd_dptable_create_with_options(const whamd_readset_view* readset, const uint32_t* recombcost,
                                                 size_t n_recombcost, const whamd_pedigree_view* pedigree,
                                                 int distrust_genotypes, const uint32_t* positions, size_t n_positions,
                                                 const char* const* keys, const char* const* values, size_t n_options,
                                                 int device, whamd_dptable** table_out);
whamd_status_t whamd_dptable_solve(whamd_dptable* table);
/* The two halves of whamd_dptable_solve, for the host-side work queue: _enqueue submits the table's launches to its
 * own HIP stream and returns; _wait blocks until the index path has arrived and evaluates the host part.  Several
 * tables (independent phasing blocks) may be in flight on one device at once -- their kernels overlap. */
whamd_status_t whamd_dptable_enqueue(whamd_dptable* table);
whamd_status_t whamd_dptable_wait(whamd_dptable* table);
/* ~PedigreeDPTable (src/pedigreedptable.cpp:40-46) */
void whamd_dptable_destroy(whamd_dptable* table);

/* Shape queries (valid after create). */
uint64_t whamd_dptable_column_count(const whamd_dptable* table);     /* ColumnIterator::get_column_count */
uint32_t whamd_dptable_individual_count(const whamd_dptable* table); /* Pedigree::size */
uint32_t whamd_dptable_read_count(const whamd_dptable* table);       /* ReadSet::size */
/* positions[column_count]: ColumnIterator::get_positions (src/columniterator.cpp:81-83) */
whamd_status_t whamd_dptable_positions(const whamd_dptable* table, uint32_t* positions_out);

/* PedigreeDPTable::get_optimal_score (src/pedigreedptable.cpp:338-341). Valid after solve. */
whamd_status_t whamd_dptable_get_optimal_score(const whamd_dptable* table, uint32_t* score_out);

/*
 * PedigreeDPTable::get_super_reads (src/pedigreedptable.cpp:344-388).  Instead of building Read
 * objects the ABI returns their contents; the binding creates, per individual i (pedigree order),
 * Read("superread_0_<i>", -1, -1, sample_id_out[i]) and Read("superread_1_<i>", ...) and adds, for
 * column c, (positions[c], allele0_out[i*n + c], quality_out[i*n + c]) resp. allele1_out.
 *   allele*_out  : [n_individuals * n_columns], values 0, 1 or 3 (EQUAL_SCORES)
 *   quality_out  : [n_individuals * n_columns]
 *   transmission_out : [n_columns]  (index_path[c].inheritance_value)
 *   sample_id_out    : [n_individuals] (Pedigree::index_to_id)
 * Any output pointer may be NULL to skip it.
 */
whamd_status_t whamd_dptable_get_super_reads(const whamd_dptable* table, uint8_t* allele0_out,
                                             uint8_t* allele1_out, uint32_t* quality_out,
                                             uint32_t* transmission_out, uint32_t* sample_id_out);

/* PedigreeDPTable::get_optimal_partitioning (src/pedigreedptable.cpp:391-406) with the Cython
 * post-processing of core.pyx:413-416 already applied: partition_out[r] in {0, 1} is the side
 * (the bit) of read r; reads never active get 1. */
whamd_status_t whamd_dptable_get_optimal_partitioning(const whamd_dptable* table, uint8_t* partition_out);

/* The raw backtrace result (index_path, src/pedigreedptable.h:17-21,54): bipartition index and
 * transmission value per column.  Not exposed by the reference's Cython layer; used by parity tests. */
whamd_status_t whamd_dptable_get_index_path(const whamd_dptable* table, uint32_t* index_out,
                                            uint32_t* transmission_out);

/* Measurements of the last whamd_dptable_solve on this table. */
whamd_status_t whamd_dptable_get_stats(const whamd_dptable* table, whamd_solve_stats* stats_out);

/* Options (for A/B measurements and tests), effective at the next solve:
 *   "path"          "auto" (default: slot runs for a single individual and for one or two trios -- pedigree slot runs, csrc/kernels_pedslots.h --,
 *                   the per-column kernels for larger pedigrees) | "slots" | "resident" (round 1's LDS-resident runs, kept for comparison) |
 *                   "column" (one launch per column, the general path) | "column_keys"
 *   "slot_l"        preferred number of local slots of a slot run (slot_r + 6 .. slot_r + 9: 1 .. 8 waves per workgroup)
 *   "slot_r"        reg slots of a slot run: "2" (4 cells per thread, default) or "3" (8 cells per thread)
 *   "arena_limit_bytes"  upper bound of the backtrace arena (default: what free HBM allows).  A table whose records need
 *                   more is solved in windows: the forward pass keeps the projection column at every window boundary and
 *                   the steps of every window but the newest are run a second time right before their records are walked
 *                   (same result, up to twice the forward time, any table length).
 *   "resident_l"    preferred log2 slice size of the run kernels
 *   "resident_fold" "0" disables folding of columns without an ending read
 *   "symmetry"      single individual: D[~x] == D[x], so a run may compute half of its workgroups only: "0" never,
 *                   "1" runs that would fill the chip (default), "2" every run with a grid read (tests)
 *   "lanes"         how many connected components of a single-individual table advance side by side (default 32, their
 *                   runs go out as batched launches; "1" solves them one after the other) */
whamd_status_t whamd_dptable_set_option(whamd_dptable* table, const char* key, const char* value);

/*
 * Host-only diagnostics (no device needed): builds the flattened problem and the forward plan exactly as
 * whamd_dptable_create would and reports how the columns are scheduled.  Used by the CPU test-suite to check
 * the planner's invariants (every column in exactly one step, runs within the LDS budget, ...).
 */
typedef struct whamd_plan_summary {   /* (the CPU plan emulators that used to be declared below live in whatshap_amd_debug.h: test-only library) */
	uint64_t n_columns;
	uint64_t n_steps;             /* launches of the forward pass (runs + per-column steps) */
	uint64_t n_runs;              /* resident runs (one launch each) */
	uint64_t n_resident_columns;  /* columns executed inside runs */
	uint64_t n_folded_columns;    /* resident columns evaluated inside their successor (no barrier of their own) */
	uint64_t n_vectorised_columns;/* resident columns on the 4-entries-per-thread path (incl. folded) */
	uint64_t max_run_columns;
	uint64_t max_workgroups;      /* largest grid of a run */
	uint64_t max_lds_bytes;       /* largest dynamic LDS request of a run */
	uint64_t backtrace_bytes;     /* size of the backtrace arena */
	uint64_t n_components;        /* connected components the device driver may run as independent jobs (single individual) */
	uint64_t n_halved_runs;       /* runs that launch only half of their workgroups (complement symmetry) */
	uint32_t max_coverage;
	uint32_t invariants_ok;       /* 1 if the internal consistency checks passed */
	uint64_t n_yform_runs;        /* slot runs that compute in Y form (one absolute difference per cell-column; slots.h) */
	uint64_t n_fact_runs;         /* pedigree slot runs on factorised lines (a trio or a quartet whose genotypes are not trusted; slots.h PSLOT_FACT, PSLOT_FACT4) */
} whamd_plan_summary;
whamd_status_t whamd_plan_summarize(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                    const whamd_pedigree_view* pedigree, int distrust_genotypes,
                                    const uint32_t* positions, size_t n_positions, const char* path,
                                    whamd_plan_summary* out);

/* ---- PedMecHeuristic: the beam-search sibling of PedigreeDPTable behind the same Python API -------------
 * Replaces cpp.PedMecHeuristic (whatshap/cpp.pxd:260-268; src/pedmecheuristic.h:56-120): constructor arguments of
 * whatshap/core.pyx:674-689 (row_limit, allow_mutations; verbosity has no counterpart), solve() is part of _create as the
 * wrapper's getters all call it first (core.pyx:695).  The ReadSet must be sorted (whatshap/cli/phase.py:590 sorts it) and
 * the pedigree's individuals must carry the ids 0 .. n-1 in insertion order (the reference mixes ids, indices and ranks,
 * src/pedmecheuristic.cpp:49-82).  One persistent workgroup per table (csrc/heuristic_device.hip), any number of tables per launch
 * (whamd_pedmec_heuristic_enqueue_many); every decision of the beam equals the reference's (float scores restated operation by operation). */
typedef struct whamd_heuristic whamd_heuristic; /* opaque */
typedef struct whamd_heuristic_stats {
	uint64_t n_columns, n_reads;
	uint64_t max_solutions;      /* widest column of the beam */
	uint64_t total_solutions;    /* sum over the columns */
	double device_ms;            /* HIP events around the kernel */
	double host_prepare_ms;      /* flattening + per-column bookkeeping + upload (wall) */
	double host_finish_ms;       /* allele votes + phasing per column (wall) */
	uint32_t n_samples, row_limit;
} whamd_heuristic_stats;
whamd_status_t whamd_pedmec_heuristic_create(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                             const whamd_pedigree_view* pedigree, int distrust_genotypes,
                                             const uint32_t* positions, size_t n_positions, uint32_t row_limit, int allow_mutations,
                                             int device, whamd_heuristic** out);
/* Several tables at once, asynchronously: every job is what whamd_pedmec_heuristic_create takes.  _enqueue_many builds the plans (a few
 * host threads), uploads them and submits ONE launch whose grid is the tables -- one persistent workgroup each, on a stream of the batch's
 * own -- and returns with out[i] in flight; whamd_pedmec_heuristic_wait(out[i]) collects (the first wait on any handle of a batch collects
 * the whole batch; the getters fail on a handle still in flight).  Independent tables are what `whatshap phase --algorithm heuristic`
 * produces per chromosome x family (whatshap/cli/phase.py:467,486,589-603).  The input arrays are only read during _enqueue_many. */
typedef struct whamd_heuristic_job {
	const whamd_readset_view* readset;
	const uint32_t* recombcost;
	size_t n_recombcost;
	const whamd_pedigree_view* pedigree;
	int distrust_genotypes;
	const uint32_t* positions;
	size_t n_positions;
	uint32_t row_limit;
	int allow_mutations;
} whamd_heuristic_job;
whamd_status_t whamd_pedmec_heuristic_enqueue_many(const whamd_heuristic_job* jobs, size_t n_jobs, int device, whamd_heuristic** out);
whamd_status_t whamd_pedmec_heuristic_wait(whamd_heuristic* h);
uint64_t whamd_pedmec_heuristic_column_count(const whamd_heuristic* h);
uint32_t whamd_pedmec_heuristic_sample_count(const whamd_heuristic* h);
uint32_t whamd_pedmec_heuristic_read_count(const whamd_heuristic* h);
/* score: getOptScore (the reference never assigns it: 0); bipartition[reads]: getOptBipartition bits; transmission[columns]:
 * getOptTransmission; haplotypes / mutated [samples][2][columns]: getOptHaplotypes / getMutations; sample_ids[samples]: the
 * global ids getSuperReads gives its reads; positions[columns].  NULL pointers are skipped. */
whamd_status_t whamd_pedmec_heuristic_get(const whamd_heuristic* h, float* score, uint8_t* bipartition, uint32_t* transmission,
                                          int8_t* haplotypes, uint8_t* mutated, uint32_t* sample_ids, uint32_t* positions);
whamd_status_t whamd_pedmec_heuristic_get_stats(const whamd_heuristic* h, whamd_heuristic_stats* stats_out);
void whamd_pedmec_heuristic_destroy(whamd_heuristic* h);

/* The tie-break hash of ReadSet::sort (src/readset.h:39-66,76-82): std::hash<std::string>(name) ^
 * std::hash<int>(source_id) of the libstdc++ this library is built against.  Used by the Python
 * mirror of ReadSet.sort(); not part of the DP path. */
uint64_t whamd_read_sort_hash(const char* name, int source_id);

/* Read selection (whatshap/readselect.pyx:218-255 readselection(readset, max_cov, preferred_source_ids, bridging)):
 * selected_out[r] = 1 for every read the reference would return, 0 otherwise.  Host code (a priority queue; SURVEY.md
 * section 8 row f2); ties are resolved as the reference resolves them under CPython 3.10 / libstdc++ (readselect.cpp).
 * read_source_id: [n_reads] Read::getSourceID, may be NULL when n_preferred == 0.  A read with fewer than two variants
 * is WHAMD_ERR_INVALID (the reference raises ValueError).  Uses var_position and var_quality of the view. */
whamd_status_t whamd_readselection(const whamd_readset_view* readset, const int32_t* read_source_id,
                                   const int32_t* preferred_source_ids, size_t n_preferred, uint32_t max_cov, int bridging,
                                   uint8_t* selected_out, uint64_t* n_selected);

/* ---- GenotypeDPTable -----------------------------------------------------------------
 * Replaces  cppclass GenotypeDPTable (whatshap/cpp.pxd:118-121; src/genotypedptable.cpp):
 *     GenotypeDPTable(ReadSet*, vector[unsigned int] recombcost, Pedigree* pedigree, vector[unsigned int]* positions)
 *     vector[long double] get_genotype_likelihoods(unsigned int individual_id, unsigned int position)
 * One call = constructor (the whole forward-backward pass) + get_genotype_likelihoods for every individual and column.
 * The pedigree view must carry genotype_likelihoods: here they are the genotype PRIORS (probabilities of 0/0, 0/1, 1/1;
 * the reference asserts they are present, src/transitionprobabilitycomputer.cpp:66); its `genotype` codes are not used.
 * gl_out: [n_individuals][n_columns][3], every triple sums to 1.  n_columns = n_positions, or the number of distinct
 * read positions when positions is NULL; gl_capacity (in doubles) must be at least n_individuals * n_columns * 3.
 * Arithmetic is f64 (the reference: long double): results agree to a relative tolerance (~1e-12), not bit for bit. */
typedef struct whamd_genotype_stats {
	uint64_t n_columns;
	uint64_t n_cells;        /* sum_c 2^k_c */
	uint64_t launches;
	double backward_ms;      /* HIP events: backward pass that leaves the kept columns */
	double forward_ms;       /* HIP events: windows (backward recompute + forward + normalisation) */
	double total_ms;
	double host_prepare_ms;  /* wall: flattening + model tables + upload + solve + download, minus total_ms */
	uint32_t window;         /* columns per window */
	uint32_t max_coverage;
	uint32_t transmissions;
	uint32_t slot_runs;      /* launches per chain of the run-fused path (genotype_slots.hip); 0: the per-column kernels ran */
} whamd_genotype_stats;
whamd_status_t whamd_genotype_likelihoods(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                          const whamd_pedigree_view* pedigree, const uint32_t* positions, size_t n_positions,
                                          int device, uint32_t window, double* gl_out, size_t gl_capacity,
                                          whamd_genotype_stats* stats_out);

/* Everything the library keeps between calls goes back to the driver: the column store of whamd_genotype_likelihoods (tens of GB for long
 * inputs, one block per device: mapping that much fresh device memory takes seconds), the backtrace arena of the table closed last
 * (commonly > 10 GB), the pinned upload staging area of whamd_dptable_create (up to 1 GiB of host memory) and the device buffers of
 * the PedMecHeuristic solves.  Blocks in use are not touched. */
void whamd_release_caches(void);
/* Host memory the library keeps idle between tables, in bytes: a table's large arrays (columns, entries, plan rows, descriptors, solution -- 45 MB for a
 * coverage-15 table of 50 000 columns, 200 MB for configs[2]) go back to a process-wide pool when the table is destroyed and are reused by the next create
 * (csrc/host_memory.cpp: freeing and re-faulting them was 6 ms per configs[2] table and 0.7 ms per small one, and serialised concurrent creates).
 * whamd_release_caches() returns them to the system; WHAMD_HOST_POOL_MB bounds the pool (default 16384, 0 switches it off).  No reference counterpart:
 * the reference frees everything with the table. */
uint64_t whamd_host_pool_idle_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* WHATSHAP_AMD_H */
