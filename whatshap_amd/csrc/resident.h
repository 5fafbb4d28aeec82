// resident.h -- the "resident" forward path: a run of consecutive columns is processed by ONE launch in which every
// workgroup keeps its slice of the projection column in LDS (single individual: kernels_resident.h; trio: kernels_trio.h).
//
// Bits of the state are split per run ("segment"): g *grid reads* -- reads that stay active through the whole run --
// index the 2^g workgroups; all other reads are *local* bits of a workgroup's LDS slice.  Reads that start inside the
// run become new (top) local bits, reads that end inside the run are minimised out of the local index.  Between two
// runs the projection column goes through HBM in an *exchange layout* chosen per boundary (next run's grid reads on
// top, then the previous run's, then the bits local in both), so the next run is free to choose other grid reads; the
// re-layout is the only time Pr touches HBM.  Columns that do not fit a run (last column, > 3 reads ending at once,
// slices larger than LDS) are executed by the per-column kernels (kernels_column.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "problem.h"

namespace whamd {

constexpr int RES_EMAX = 3;        // reads that may end in one resident column
constexpr int RES_LMAX = 13;       // log2 of the largest LDS slice (entries)
constexpr int RES_GMAX = 10;       // log2 of the largest grid (workgroups) of one run (coverage 23 = 10 grid + 13 local bits)
constexpr int RES_MAXCOLS = 48;    // columns per run (bounds the LDS lookup tables)
constexpr int RES_TABLE = 256;     // per column: two 128-entry tables (low / high 7 local bits)

// Self-contained descriptor of one resident column (320 B); a run copies the leading RES_LDSWORDS of its descriptors
// to LDS at kernel start so that no per-column global (HBM-latency) load sits on the sequential column chain.
// Per-column constants reach the lanes as 16-byte LDS broadcast reads -- the CU has ONE scalar ALU shared by all waves,
// so per-column scalar work replicated per wave must be avoided -- but a broadcast read still occupies the LDS pipe
// for the full 64 x 16 bytes (8 cycles): with 16 waves the 13 reads per step of the first layout were ~1 700 of the
// ~2 000 cycles of a step.  The words the packed evaluation needs are therefore packed into Q0..Q3 (4 reads for the
// column, 2 per folded column).
constexpr int RES_LDSWORDS = 64;
struct ResColumn {
	// ---- Q0..Q3 (words 0..15): everything the packed 16-bit evaluation of a vectorised column reads
	uint32_t A;                       // Cp from the host; the kernel adds S_grid of its workgroup (A + table entries = Cp + S)
	uint32_t Kpk, Ccpk;               // (Cp + Cm) and min(Cc, 0xFFFF), each in both 16-bit halves
	uint32_t flags;                   // mode | nfold << 8 | pk_ok << 12
	uint32_t pk[4];                   // subset sums of d0, d1, d2 for the cell pairs (0,1) (2,3) (4,5) (6,7): low | high << 16
	uint32_t lowmask, nthr, stage_off, nwords;  // 2^Lb - 1; threads of the vectorised path (2^Lf / 4); word offset of this
	                                  // column's record in the workgroup's staging area; words per plane
	uint32_t ep0, mL0, PGq, pbits;    // copies of epos[0], mL[0], PG (kernel); pbits: bit u = parity of (low cell bits of
	                                  // entry 4t+u's side-0 cell) & mL0 -- the part of the tie-break parity that is the same for every thread;
	                                  // bit 8 = parity of the number of cell bits above the ending read (mirror-image tie rule)
	// ---- words 16..35: the 32-bit paths (vectorised without pk_ok, generic)
	uint32_t Cp, Cm, Cc, mode;        // cost = min(Cp + S, Cm - S, Cc); an absent Cp / Cm is RES_ABSENT (never the minimum:
	                                  // the planner guarantees every real value < 2^30).  mode: RES_MODE_*
	uint32_t epos[4];                 // local bit position of ending read q (ascending logical position); [3] unused
	uint32_t mL[4];                   // per ending read: local cell bits logically above it (tie-break parity); [3] unused
	int32_t Sg;                       // written by the kernel: sum of the grid-read deltas selected by the workgroup index
	uint32_t PG;                      // written by the kernel: bit q = parity of the workgroup-index bits above ending read q
	uint32_t Lb, Lf;
	int32_t d0, d1, d2, dE;           // deltas of local cell bits 0, 1, 2 and of ending read 0
	// ---- cold
	uint32_t ebits, nfold;            // nfold: preceding columns folded into this one (they have mode RES_MODE_FOLDED)
	int32_t dloc[14];                 // signed deltas of the local bits
	uint32_t pk_ok;                   // Cp + Cm < 2^14 for this column and every column folded into it: four column costs
	                                  // still add up in 16 bits
	uint32_t pad1[11];
	// ---- global only (words 64..79): read once per run for the workgroup-dependent scalars
	uint32_t mG[4];                   // per ending read: grid bits logically above it
	int32_t dgrid[RES_GMAX];          // signed deltas of the grid reads at this column
	uint32_t pad2[2];
};
static_assert(offsetof(ResColumn, mG) == RES_LDSWORDS * 4, "LDS part of ResColumn");
static_assert(offsetof(ResColumn, Cp) == 64 && offsetof(ResColumn, d0) == 128, "32-bit hot words of ResColumn");
static_assert(sizeof(ResColumn) == 320, "ResColumn must stay 80 words");
constexpr uint32_t RES_ABSENT = 0xC0000000u;
// vectorised modes: a thread owns 4 consecutive projection entries and moves them with 16-byte LDS accesses
constexpr uint32_t RES_MODE_E0 = 0;       // no read ends
constexpr uint32_t RES_MODE_E1_HIGH = 1;  // one read ends, local bit >= 2
constexpr uint32_t RES_MODE_E1_BIT0 = 2;  // one read ends, local bit 0
constexpr uint32_t RES_MODE_E1_BIT1 = 3;  // one read ends, local bit 1
constexpr uint32_t RES_MODE_GENERIC = 4;  // anything else (<= 3 reads ending, tiny slices)
constexpr uint32_t RES_MODE_FOLDED = 5;   // no read ends here and the next column is vectorised: this column's cost is added
                                          // inside the next column's evaluation (no slice traffic, no barrier of its own)
constexpr uint32_t RES_MAXFOLD = 3;

// ---- trio runs (T = 4, three individuals): every slice entry is a vector of T values; a thread evaluates one
// projection entry (all T transmission values, all cells projecting onto it).  Descriptor of one column, 512 bytes.
constexpr int PED_T = 4, PED_NIND = 3;
constexpr int PED_LMAX = 10;       // log2 of the largest slice (entries of T values)
constexpr int PED_LKMAX = 12;      // local cell bits (two 6-bit lookup tables per individual)
constexpr int PED_TABLE = PED_NIND * 128;  // table words per column
// One cost term of a transmission value: c + sum_s sig_s * L_s, sig_s in {-1, 0, +1} as signed byte s of `sig`
// (restates the plus / minus individual masks of CostTerm).  |L_s| < 2^23 is a planner guarantee (24-bit multiply-add).
struct PedTerm { uint32_t c, sig; };
constexpr int PED_REGTERMS = 4;    // terms per transmission value held in the descriptor (and in registers)
constexpr int PED_LDSWORDS = 80;   // leading words of PedColumn a run copies into LDS
struct PedColumn {
	// ---- LDS part: every address below depends only on (column, lane), so one LDS latency covers all of it
	uint32_t Lb, Lf, ebits, stage_off;   // stage_off: u32 words into the workgroup's record (one u32 per projection entry)
	uint32_t lowmask, recomb, n_terms, term_off;  // term_off: index into the run's term pool
	uint32_t epos[4], mL[4];             // ending reads (ascending logical position): local position, tie-break mask
	uint32_t PG, maxcnt, pad0[2];        // PG written by the kernel; maxcnt: most terms any transmission value has
	uint32_t tptr[8];                    // [T + 1] term ranges per transmission value, relative to term_off
	int32_t Sg[4];                       // per individual, written by the kernel
	int32_t dE[RES_EMAX][4];             // delta of ending read q for individual s (non-zero for its own individual only)
	uint32_t pad1[4];
	PedTerm rterms[PED_T][PED_REGTERMS]; // first terms of every transmission value, padded with {INF, 0}
	// ---- global only
	uint32_t mG[4];                      // grid part of the tie-break masks
	int32_t dgrid[PED_NIND][RES_GMAX];   // signed deltas of the grid reads, per individual (0 for reads of other individuals)
	int32_t dloc[PED_NIND][PED_LKMAX];   // signed deltas of the local bits, per individual
	uint32_t pad2[10];
};
static_assert(sizeof(PedColumn) == 640, "PedColumn must stay 160 words");
static_assert(offsetof(PedColumn, mG) == PED_LDSWORDS * 4, "LDS part of PedColumn");

// Passed to the kernel by value (kernel arguments live in SGPRs: no memory round trip before the first column).
constexpr int RES_IOSEG = 6;       // runs per mask of the load / store layouts held in the kernel arguments
struct ResSegment {
	uint32_t c0, ncols, g, col_off;
	uint32_t Lb0, Lf_last, has_prev, threads;
	uint32_t max_l, pad;
	uint16_t bt_active, bt_simple;  // backtrace chain: columns in which a read ends; 1: all of them use one-byte-per-thread
	                                // records (single individual); 2: trio run with at most one ending read per column
	uint32_t kind;         // 0: single individual (resident_segment), 1: trio (resident_segment_ped)
	uint32_t term_off, n_terms;  // trio: this run's slice of the term pool
	uint32_t stage_words;  // ballot words (u64) one workgroup produces in this run
	uint32_t bt_lo, bt_hi; // byte offset of the run's backtrace record: [workgroup][stage_words] u64
	uint32_t n_wext;       // runs extracting the workgroup index from the logical exit index
	uint32_t wext[RES_IOSEG];
	uint32_t n_lext;       // runs extracting the local exit index from the logical exit index
	uint32_t lext[10];
	uint16_t n_in_grid, n_in_local, n_out_grid, n_out_local;
	// packed runs (compact position | mask position << 8 | length << 16): logical index = OR of deposits of w and l
	uint32_t in_grid[RES_IOSEG], in_local[RES_IOSEG], out_grid[RES_IOSEG], out_local[RES_IOSEG];
	// Complement symmetry (single individual: D[~x] = D[x]): a run with `half` set launches only
	// the workgroups whose top grid-read bit is 0; the other half of every column is its mirror image.
	uint32_t half;          // 1: launch 2^(g-1) workgroups
	uint32_t mirror_out;    // 1: also store the mirror image of the exit slice (the next step reads every entry)
	uint32_t in_half;       // 1: the entering slice was written by a halved run without mirror_out: entries whose bit
	uint32_t in_mirror_bit; //    `in_mirror_bit` is set are read from their complement (index ^ in_fullmask)
	uint32_t in_fullmask, out_fullmask;  // all index bits of the entering / exit slice
};

// One run of a batched launch (resident_batch): the segment and the buffers of its job's lane.
struct ResBatchEntry {
	ResSegment sg;
	const uint32_t* prev;
	uint32_t* cur;
	uint32_t* score_out;  // non-null: the run ends a connected component, its single exit value goes here
};
static_assert(sizeof(ResBatchEntry) % 8 == 0, "entries hold pointers");

// Everything the backtrace needs for one resident column, self-contained (128 B) so that a run's records can be staged
// in LDS with one coalesced copy.  The walk stays in the run's LOCAL index space (the grid-read bits of the path are
// constant inside a run): with cell_{c+1} the local cell index of the path at column c+1,
//   l_out(c) = cell_{c+1} & (2^Lf - 1),  cell_c = l_out with zeros inserted at epos[] and the recorded argmin bits OR-ed in.
// Only columns in which a read ends cost an LDS access on the sequential chain.  The logical bipartition index
// x_c = deposit(w, gruns) | deposit(cell_c, lruns) is produced afterwards, one lane per column.
constexpr int RES_BT_GRUNS = 8, RES_BT_LRUNS = 10;
struct ResBacktrace {
	uint32_t Lf, ebits, layout, stage_off;  // layout 0: ballot planes (word l >> 6, bit l & 63);  1: one byte per thread l >> 2, bit l & 3;
	                                        // 2: trio, one byte per (entry, transmission value): ending-read bits | argj << 3
	uint32_t nwords, epos[3];               // local bit positions of the ending reads (ascending logical position)
	uint32_t n_g, n_l;                      // runs in use
	uint32_t gruns[RES_BT_GRUNS];           // workgroup-index bits -> logical positions (source | destination << 8 | length << 16)
	uint32_t lruns[RES_BT_LRUNS];           // local cell bits      -> logical positions
	// The sequential chain visits only the columns in which a read ends ("active", in descending column order); a
	// column without an ending read is derived afterwards: cell = (cell of the next active column above it, or the
	// run's exit index) & cmask.
	uint32_t kpos;                          // position of this column in the chain, RES_BT_NONE if no read ends here
	uint32_t src;                           // column (in the run) of the next active column above, RES_BT_NONE: the exit index
	uint32_t cmask;                         // AND of (2^Lf - 1) from this column up to (excluding) column `src`
	uint32_t kcol;                          // record k: column (in the run) of chain position k
};
constexpr uint32_t RES_BT_NONE = 0xFFFFFFFFu;
static_assert(sizeof(ResBacktrace) == 128, "ResBacktrace must stay 32 words");

// Backtrace unit list (reverse processing order): what the backtrace needs to know about a step without chasing
// pointers, 128 bytes, so that headers can be prefetched two units ahead.
struct BtUnit {
	uint32_t kind;         // 0 = column step, 1 = resident run
	uint32_t c0, ncols;    // first column / number of columns
	uint32_t col_off;      // run: index of its first record in the ResBacktrace array
	uint32_t g, Lf_last, stage_words, n_wext;
	uint32_t bt_lo, bt_hi, n_lext, pad0;   // run: pad0 = active columns | simple << 16 | half << 20 (ResSegment bt_active / bt_simple / half)
	uint32_t wext[RES_IOSEG];  // logical exit index -> workgroup index
	uint32_t lext[RES_BT_LRUNS];  // logical exit index -> local exit index
	uint32_t pad1[4];
};
static_assert(sizeof(BtUnit) == 128, "BtUnit must stay 32 words");

// One backtrace job (blockIdx.x of backtrace_kernel): a contiguous range of units, newest first.
struct BtJob {
	uint32_t unit_off, unit_count;
	uint32_t with_last_column;  // 1: units[0] is the table's last column (optimum from the key scratch); 0: start at entry 0;
	                            // 2: start from DevProblem::bt_state (windowed solve)
	uint32_t pad;
};

struct Step {
	uint32_t kind;         // 0 = one column through the column kernels, 1 = resident run
	uint32_t index;        // column index, or index into segments
};

struct ResidentPlan {
	std::vector<Step> steps;
	std::vector<ResSegment> segments;
	std::vector<ResColumn> columns;      // resident columns in run order
	std::vector<int32_t> col_to_res;     // [n_cols] index into `columns` or -1
	std::vector<ResBacktrace> backtrace; // parallel to `columns`
	std::vector<PedColumn> ped_columns;  // parallel to `columns` (filled for trio runs only)
	std::vector<PedTerm> ped_terms;      // term pool of the trio runs
	uint64_t n_resident_columns = 0;
	// Single individual: indices into `steps` at which a new connected component of the ReadSet starts (no read is active
	// across the boundary, b == 0; always contains 0).  Components are independent sub-problems whose optimal costs add
	// up (whatshap_amd/blocks.py has the exactness argument); runs never span such a boundary.
	std::vector<uint32_t> component_first_step;
};

// Plans the whole forward pass.  `resident` false -> every step is a per-column step.
// use_symmetry: 0 never halve runs, 1 halve full-chip runs (>= 2^8 workgroups), 2 halve every run with a grid read (tests)
void plan_forward(const Problem& p, bool resident, int l_pref, bool fold, ResidentPlan& plan, int use_symmetry = 1);

}  // namespace whamd
