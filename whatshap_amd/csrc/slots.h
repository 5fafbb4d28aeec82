// slots.h -- "slot runs": the register-resident forward path of a single individual (T = 1), kernels_slots.h.
//
// The projection column of a run never leaves the register file.  Every read that is active somewhere in the run owns
// one *slot* = one bit of a PHYSICAL cell index P = (workgroup << L) | (wave << (6 + LR)) | (lane << LR) | reg:
//
//     reg slots   (LR bits)  the R = 2^LR cells a thread keeps in registers differ in these reads
//     lane slots  (6 bits)   the 64 lanes of a wavefront
//     wave slots  (LW bits)  the waves of the workgroup
//     grid slots  (g bits)   the workgroups of the launch (reads that stay active through the whole run)
//
// A column adds cost(x) to every cell (closed form, DESIGN.md section 2: min3(Cp + S, Cm - S, Cc) with S the sum of the
// deltas of the set bits); a read that ENDS is minimised out by combining the two cells that differ in its slot --
// in registers (reg slot), with one cross-lane move (lane slot) or through a small LDS exchange (wave slot) -- after
// which both cells hold the minimum: the slot is *free* (its two halves are duplicates) until a read that STARTS takes
// it over.  Nothing is compacted or re-indexed between columns, so a column without an ending read costs no
// communication at all, and a column with one costs no workgroup barrier unless the read sits in a wave slot.
// Grid-slot reads cannot be minimised inside a launch: the run ends before the first of them does, the column goes
// through HBM in the NEXT run's physical order (the writer scatters, the reader loads contiguously) and the next run
// picks new grid reads.  (The LDS-resident runs of resident.h re-index the slice at every ending read: one LDS round
// trip and one workgroup barrier per column step, which is what bounded them at ~13 % of the VALU issue rate.)
//
// Tie-breaking (src/pedigreedptable.cpp:306-327): of two cells that differ in ending read h the one whose
// h-bit equals the parity of the bits of reads logically ABOVE h wins a tie.  Every cell stores one decision bit per
// ending read: the side-0 cell of a pair the pair's decision, the side-1 cell the decision of the pair's MIRROR IMAGE
// (all bits complemented) -- a halved run (complement symmetry, D[~x] = D[x]) computes only the workgroups whose top
// grid-slot bit is 0, and the backtrace reads the mirror decisions where the path runs through the other half.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "problem.h"
#include "host_parallel.h"
#include "resident.h"

namespace whamd {

constexpr int SLOT_LR = 3;           // most reg slots of a run (8 cells per thread); SlotRun::lr says how many a run uses (2 or 3)
constexpr int SLOT_LANE = 6;         // lane slots
constexpr int SLOT_LWMAX = 3;        // wave slots (at most 8 waves per workgroup)
constexpr int SLOT_GMAX = 12;        // grid slots
constexpr int SLOT_MAXSLOTS = 26;    // local + grid slots of one run (device limit: 25 reads per column)
constexpr int SLOT_MAXCOLS = 64;     // columns per run: lane c of every wave prepares column c in the prologue
constexpr int SLOT_MAXEND = 8;       // reads that may end in one column of a run (pedigree runs: PSLOT_MAXEND)
constexpr int PSLOT_MAXEND = 4;      // ... of a pedigree run (four decision bits next to the transmission argument in a lane's record byte)
constexpr int SLOT_MAXENDS_RUN = 96; // ending reads per run (one record byte per thread each)
constexpr int32_t SLOT_DELTA_LIMIT = 1 << 22;  // |delta| of every read in a run (24-bit multiply-add of the lane part)

// Per-column descriptor of a run, 64 dwords.  The first 16 ("hot") are what a run reads per column: the prologue copies the
// hot lines of its columns into LDS (kernels_slots.h).  The rest ("cold") is read once per table, by slot_tables.
struct SlotRow {
	// ---- hot: ONE 64-byte line per column
	// (rows of a Y-form run -- SlotRun::yflags -- hold Kr[0..3] in the first four words instead: Kr[r] = K - 2 * (sum of the reg-slot
	//  deltas set in r) + SLOT_YBIAS; the planner rewrites them once the runs are known, see slot_plan.cpp "Y form")
	uint32_t K;                      // Cp + Cm (mod 2^32; an absent term is RES_ABSENT, resident.h)
	uint32_t Cc;                     // constant term (INF if none)
	int32_t dreg[SLOT_LR];           // deltas of the reg slots (0 for a slot the run does not have)
	int32_t dlane[SLOT_LANE];        // deltas of the lane slots
	uint32_t n_end;                  // reads ending in this column (<= SLOT_MAXEND)
	struct End {
		uint32_t info;               // slot | qmask << 8 | mflip << 24 (qmask bit r: reg-slot part of the tie-break parity of cell r,
		                             //  including (side & mflip) when the ending read itself sits in a reg slot)
		uint32_t M;                  // physical index bits of the reads logically above the ending read
	} end[SLOT_MAXEND];              // ascending logical position; end[2] and later lie in the next line (three and more reads ending at
	                                 // once: rare in a regular layout, every tenth run of an irregular one -- the kernel fetches them with scalar loads)
	uint32_t pad2[4];
	// ---- cold (128-byte aligned: lane c fetches its column with wide loads)
	int32_t dslot[SLOT_MAXSLOTS];    // delta of every slot at this column (0: free slot, BLANK entry)
	uint32_t Cp;
	uint32_t pad3[5];
};
static_assert(SLOT_LR == 3 && SLOT_LANE == 6 && SLOT_MAXEND == 8, "hot layout of SlotRow: 2 + 3 + 6 + 1 + 4 dwords in the first line, six more ending reads in the second");
static_assert(sizeof(SlotRow) == 256, "SlotRow must stay 64 dwords");
constexpr int SLOT_CTRL_WORDS = 32;  // control words of a run: single individual 16 bits per column (n_end | slot of the first ending read << 2 | its
                                     // qmask << 7), SLOT_MAXCOLS columns; pedigree runs one byte per column (n_end | slot << 2), PSLOT_MAXCOLS columns
constexpr uint32_t SLOT_YBIAS = 0x80000000u;   // Y-form rows and tables: both operands of the absolute difference carry this bias (unsigned compare)
constexpr int SLOT_XCOLS = 32;      // X runs: columns whose per-thread operand stays in registers for the whole run (slot_runx_body)
constexpr int SLOT_XENDS = 32;      // X runs: ending reads of a run (one tie-parity bit per thread each, one word)
constexpr int SLOT_XPAD = 8;        // X runs: columns of slack behind the Kr table (the words of the next trip are requested a trip ahead)
constexpr int SLOT_ROW_PAD = 64;     // rows appended to the array: the kernel touches a fixed number of rows to warm the scalar cache

// One run, passed to the kernel by value.
struct SlotRun {
	uint32_t c0, ncols, g, L;        // L = lr + 6 + lw local slots
	uint32_t lw, half, has_prev, row_off;   // half: launch 2^(g-1) workgroups (top grid slot = 0); row_off: first SlotRow of the run
	uint32_t n_ends, rec_lo, rec_hi, threads;  // record: [workgroup][ending read][thread] bytes at this offset of the arena
	uint32_t in_occ, in_identity, in_half, in_mirror_pos;  // entry: occupied slots; 1: entry index == P & in_occ; the entering column was
	                                 // written by a halved run: entries whose bit in_mirror_pos is set are read at index ^ in_fullmask
	uint32_t in_fullmask, out_occ, mirror_out, out_fullmask;  // exit: occupied slots; 1: also store the mirror image (index ^ out_fullmask)
	uint32_t spec_id, lr, ctrl_off, pad;   // spec_id != 0: this run ends a backtrace chunk -- the kernel leaves (min value, index) of
	                                 // its exit column in spec_keys[spec_id - 1] (kernels_backtrace.h, speculative walk); lr: reg slots of this run (cells per thread = 2^lr); ctrl_off: the run's first word in slot_ctrl
	uint32_t in_pos[8];              // entry index bit of every occupied slot, one byte each (when !in_identity)
	uint32_t out_pos[8];             // exit index bit of every occupied slot, one byte each
	// (words, not byte arrays: the kernel reads them with static indices out of SGPRs; see slot_pos / slot_set_pos)
	// Single-individual runs: what the prologue used to compute per launch from the cold part of SlotRow is a table built once
	// at create time (slot_tables): G [launched workgroups][ncols] = Cp + deltas of the set grid slots, W [waves][ncols] = deltas
	// of the set wave slots, SL [ncols][64] = deltas of the set lane slots.  Word offsets into DevProblem::slot_tab.
	uint32_t tab_g, tab_w, tab_sl, tab_kr;   // tab_kr: X runs (below) -- the Kr words of the run's columns, contiguous: [ncols + SLOT_XPAD][2^lr]
	// Y-form runs (kernels_slots.h, "Y form"): yflags bit 0 the run computes in Y form, bit 1 the entering column is already in Y form,
	// bit 2 the exit column stays in Y form; base_in / base_out: the column-uniform base B at the run's entry and exit.
	// bit 3: the run is eligible for the register-resident X kernel (slot_runx: Y form, <= SLOT_XCOLS columns, <= SLOT_XENDS ending reads);
	// tab_par: its tie-parity table -- [threads] words (bit e: parity of the thread's local index under the mask of the run's e-th ending read)
	// followed by [launched workgroups] words (the same for the workgroup's grid bits).
	uint32_t yflags, base_in, base_out, tab_par;   // (yflags bit 4, pedigree runs: the min-plus step on packed keys value << TB | j -- kernels_pedslots.h)
};
// Dynamic LDS of a single-individual run: wave-slot exchange 2 x [threads][cells] | hot lines [ncols + 8][16] | A [8 waves][64] | lane sums
// [ncols + 8][64] (eight lines of slack: lines are requested up to six columns ahead).  Sized by the run's own length -- at 22 columns
// 28 KB instead of the 41 KB of SLOT_MAXCOLS columns -- so that FOUR workgroups of 512 threads fit a CU (160 KB) when several tables
// share a launch (slot_group).
inline size_t slot_run_lds_bytes(uint32_t threads, uint32_t lr, uint32_t ncols) {
	return (size_t)2 * threads * ((size_t)1 << lr) * 4 + (size_t)(ncols + 8) * 64 + 8 * 64 * 4 + (size_t)(ncols + 8) * 64 * 4;
}
// Dynamic LDS of an X run (slot_runx_body): wave-slot exchange 2 x [threads][4] | the threads' own operand lines [trips + 3][threads][4] (the line of
// the next trip is requested a trip ahead, two trips per loop iteration)
inline size_t slotx_lds_bytes(uint32_t threads, uint32_t ncols) { return (size_t)(2 + (ncols + 3) / 4 + 3) * threads * 16; }
inline uint32_t slot_pos(const uint32_t (&w)[8], uint32_t s) { return (w[s >> 2] >> ((s & 3u) * 8u)) & 255u; }
inline void slot_set_pos(uint32_t (&w)[8], uint32_t s, uint32_t pos) {
	w[s >> 2] = (w[s >> 2] & ~(255u << ((s & 3u) * 8u))) | (pos << ((s & 3u) * 8u));
}
static_assert(sizeof(SlotRun) == 192, "SlotRun layout");

// ---- pedigree slot runs (T = 4 or 16 transmission values; kernels_pedslots.h) -------------------------------------
// Same slots, same run / exchange machinery; a cell holds T values, ONE (cell, transmission value) per lane: the
// transmission value in lane bits 0 .. TB-1 (TB = 2 trio, 4 quartet), 6 - TB lane slots, up to 3 wave slots, no reg slots.
// The min-plus step over the previous transmission value j (src/pedigreedptable.cpp:264-300) is a butterfly of DPP moves
// over the TB low lane bits (popcount(i ^ j) * recomb is separable per bit; low bits first gives the lowest j on ties).
// The cost of a cell is the minimum over at most NF *forms* per transmission value (one per allele assignment that
// survives, PedigreeColumnCostComputer::get_cost): form a = c_a + sum over the set bits of x of sigma_a(individual) * d_bit.
// Every form splits over the slot classes, and the three parts are TABLES computed once per table at create time
// (pedslot_tables, full-chip width): G[workgroup] (constant + grid slots), W[wave], S[lane]; a run's prologue is then
// pure copies (A = G + W per wave into LDS) and a column costs two small LDS reads, NF adds and NF - 1 minima per lane.
constexpr int PSLOT_MAXCOLS = 32;      // columns per run
constexpr int PSLOT_MAXFORMS = 16;     // forms per transmission value a run can hold: NF = 2, 4, or 16 (a trio with untrusted genotypes:
                                       // 16 allele assignments, up to 15 distinct forms per value once the genotype likelihoods differ)
constexpr int PSLOT_FORMWORDS = 1024;  // ncols * T * NF of one run (one row per wave in LDS)
// NF = PSLOT_FACT: the FACTORISED line of a trio with untrusted genotypes (Problem::fterms): three signed sums {X_L, Y_L, C_L, 0} -- split
// over workgroup / wave / lane like a form: G, W, S with four words per entry -- and twelve constants {kx[4], ky[4], cc[4]} per (column,
// transmission value), the same for every cell: a fourth table K [ncols][T][12] behind S, staged once per workgroup.
//   M_a = min(kx[2a], kx[2a+1] +- X_L),  F_b = min(ky[2b], ky[2b+1] +- Y_L),  cost = min over (a, b) of cc[2a+b] (+- C_L) + M_a + F_b
// 19 operations per cell instead of 16 adds and 15 minima; a quarter of the per-wave and lane tables.
constexpr int PSLOT_FACT = 1;
constexpr uint32_t PSLOT_NK = 12;   // constants per (column, value) of a factorised line
// NF = PSLOT_FACT4: the factorised line of a QUARTET (two children of the same two founders, T = 16) with untrusted genotypes (Problem::fterms, fterm_kind 2).
// In haplotype space nothing depends on the transmission value: four signed sums {L_X, L_Y, L_C1, L_C2} and sixteen constants k[4 i + (h0 h1)] -- the cost of
// individual i = X, Y, C1, C2 carrying alleles (h0, h1): 00, 01 (+ L_i), 10 (- L_i), 11 -- per COLUMN; the tables G, W, A hold four words per column (not per
// value), K sixteen.  The transmission value only WIRES the children to the founders' haplotypes: child k carries X's haplotype u_k and Y's haplotype v_k, and
// a lane reads four predicates of its value from two words of PedSlotExtra (u_1, v_1, u_1 == u_2, v_1 == v_2).  pslot_fact4_cost below is the minimum over the
// sixteen allele assignments by elimination (55 operations; 16 forms would be 16 words per (column, value, lane) -- four columns per run).
constexpr int PSLOT_FACT4 = 3;
constexpr uint32_t PSLOT_NK4 = 16;  // constants per COLUMN of a quartet's factorised line
constexpr uint32_t PSLOT_FSTRIDE4 = 20;   // Problem::fterms entries per column (fterm_kind 2): 4 signed sums + 16 constants
constexpr bool pslot_is_fact(uint32_t nf) { return nf == (uint32_t)PSLOT_FACT || nf == (uint32_t)PSLOT_FACT4; }
constexpr uint32_t pslot_na(uint32_t nf) { return pslot_is_fact(nf) ? 4u : nf; }    // words per (column, value) of G / W / A
constexpr uint32_t pslot_ns(uint32_t nf) { return pslot_is_fact(nf) ? 4u : nf; }    // words per (column, lane) of S
constexpr uint32_t pslot_nk(uint32_t nf) { return nf == (uint32_t)PSLOT_FACT ? PSLOT_NK : (nf == (uint32_t)PSLOT_FACT4 ? 1u : 0u); }   // words per (column, value) of K (PSLOT_FACT4: T * 1 = 16 per column)
constexpr uint32_t pslot_ta(uint32_t nf, uint32_t T) { return nf == (uint32_t)PSLOT_FACT4 ? 1u : T; }   // values per column in the tables G / W / A
// The cost of one cell of a quartet whose genotypes are not trusted (shared by the kernel, kernels_pedslots.h, and the CPU emulation of the plan).
//   cost = min over the founders' alleles (a0, a1) of X, (b0, b1) of Y of  Mo[a0 a1] + Fa[b0 b1] + C1[a_u1, b_v1] + C2[a_u2, b_v2]
// In "transmitted" coordinates of child 1 -- a = a_u1, a' = X's other allele, b = b_v1, b' -- the founders' mixed entries swap with u_1 / v_1, child 1 reads
// (a, b) and child 2 reads (a or a', b or b').  Elimination: Z_q[b] = min over b' of Fa'[b b'] + C2[q, b or b'];  H[p][q] = min over b of C1[p b] + Z_q[b];
// cost = min over (a, a') of Mo'[a a'] + H[a][a or a'].  Every entry is a true cost (non-negative, below 2^30: Problem::value_bound), BIG only ever loses.
#if defined(__HIPCC__)
#define WHAMD_HD __host__ __device__
#else
#define WHAMD_HD
#endif
WHAMD_HD inline uint32_t pslot_min2(uint32_t a, uint32_t b) { return a < b ? a : b; }
WHAMD_HD inline uint32_t pslot_fact4_cost(uint32_t LX, uint32_t LY, uint32_t L1, uint32_t L2, const uint32_t (&k)[16], bool u1, bool v1, bool same_u, bool same_v) {
	constexpr uint32_t BIG = 0x40000000u;
	const uint32_t mo01 = k[1] + LX, mo10 = k[2] - LX, fa01 = k[5] + LY, fa10 = k[6] - LY;
	const uint32_t c1[2][2] = {{k[8], k[9] + L1}, {k[10] - L1, k[11]}}, c2[2][2] = {{k[12], k[13] + L2}, {k[14] - L2, k[15]}};
	const uint32_t m01 = u1 ? mo10 : mo01, m10 = u1 ? mo01 : mo10;   // Mo'[a a']
	const uint32_t f01 = v1 ? fa10 : fa01, f10 = v1 ? fa01 : fa10;   // Fa'[b b']
	// child 2 reads b: the other allele b' is free -- Fa' collapses to its row minima on the diagonal
	const uint32_t F0 = pslot_min2(k[4], f01), F1 = pslot_min2(f10, k[7]);
	const uint32_t fs00 = same_v ? F0 : k[4], fs01 = same_v ? BIG : f01, fs10 = same_v ? BIG : f10, fs11 = same_v ? F1 : k[7];
	uint32_t H[2][2];
	for (int q = 0; q < 2; ++q) {
		const uint32_t z0 = pslot_min2(fs00 + c2[q][0], fs01 + c2[q][1]), z1 = pslot_min2(fs10 + c2[q][0], fs11 + c2[q][1]);
		for (int p = 0; p < 2; ++p) H[p][q] = pslot_min2(c1[p][0] + z0, c1[p][1] + z1);
	}
	const uint32_t h01 = same_u ? H[0][0] : H[0][1], h10 = same_u ? H[1][1] : H[1][0];
	return pslot_min2(pslot_min2(k[0] + H[0][0], k[3] + H[1][1]), pslot_min2(m01 + h01, m10 + h10));
}
struct PedSlotRow {
	// ---- hot: copied to LDS by the run kernel (8 words)
	uint32_t recomb;
	uint32_t M0;                     // first ending read: physical CELL-index bits of the reads logically above it
	uint32_t info1, M1, info2, M2;   // further ending reads (rare): slot in the low byte of info
	uint32_t n_end, info0;
	// ---- cold: read by pedslot_tables only
	int32_t dslot[SLOT_MAXSLOTS];    // delta of the read in every slot at this column (0: free slot, BLANK entry)
	uint8_t ind[SLOT_MAXSLOTS + 2];  // individual of the read in every slot
	uint32_t pad[7];                 // genotype mode: [0] reads starting here, [1] starts before it in the run; phasing: [2], [3] = info3, M3 (a
	                                 // fourth ending read: fetched from the row with scalar loads, its slot for the backtrace in SlotBtCol::pad[1])
};
static_assert(sizeof(PedSlotRow) == 192, "PedSlotRow must stay 48 words");
// Per run, next to its SlotRun (kernel argument by value).
struct PedSlotExtra {
	uint32_t tb, nf, fwn, arow;      // log2 T; forms per value (or PSLOT_FACT / PSLOT_FACT4); ncols * pslot_ta(nf, T) * pslot_na(nf); fwn rounded up to 4 (row stride of A in LDS)
	uint32_t g_lo, g_hi, w_off, s_off;   // word offsets into the table array: G [2^g][fwn]; W and S relative to G: [2^lw][fwn], [ncols][64][pslot_ns(nf)] (then K [ncols][T][pslot_nk(nf)])
	uint32_t rec_words, x_off, pad[2];   // record of one workgroup: one byte per thread and column, 4 columns per word: ceil(ncols / 4) * threads words
	                                 // pad (PSLOT_FACT4): bit t of the 16-bit masks u_1 | v_1 << 16 and (u_1 == u_2) | (v_1 == v_2) << 16 (Problem::fact4_roles)
	                                 // x_off (SlotRun::yflags bit 3, pedslot_runx): relative to G like s_off -- recombination cost [ncols + SLOT_XPAD] (all-ones behind the
	                                 // run: such a column changes nothing) | control word [ncols + SLOT_XPAD] (n_end | four fields of (slot | exchange buffer << 3) from bit 3) |
	                                 // tie parities: [threads] words (bit e: parity of the lane's local cell index under the mask of the run's e-th ending read), [2^g] words
};
constexpr uint64_t pslotx_words(uint32_t ncols, uint32_t threads, uint32_t g) { return (uint64_t)2 * (ncols + SLOT_XPAD) + threads + ((uint64_t)1 << g); }
static_assert(sizeof(PedSlotExtra) == 48, "PedSlotExtra layout");
// LDS of one pedigree run: wave-slot exchange 2 x [threads] | hot lines | A [waves][arow] | S [ncols][64][ns] | K [ncols][T][nk]  (four columns
// of slack behind the rows of A, S and K: lines are requested ahead)
inline size_t pedslot_lds_bytes(uint32_t threads, uint32_t ncols, const PedSlotExtra& ex) {
	return ((size_t)2 * threads + (PSLOT_MAXCOLS + 4) * 8 + (size_t)(threads >> 6) * (ex.arow + (4u << ex.tb) * pslot_na(ex.nf)) + (size_t)(ncols + 4) * 64 * pslot_ns(ex.nf) +
	        ((size_t)(ncols + 4) << ex.tb) * pslot_nk(ex.nf)) * 4;
}

// One run as a record in device memory: what slot_batch (the runs of ONE table's jobs) and slot_group / pedslot_group (the runs of
// SEVERAL tables, one launch per super-step of the whole group) read instead of kernel arguments.  The entry carries the owning table's
// arrays, so that a launch does not need a per-table DevProblem.
struct SlotBatchEntry {
	SlotRun run;
	PedSlotExtra ex;                 // pedigree runs (zero otherwise)
	const uint32_t* prev;
	uint32_t* cur;
	uint32_t* score_out;             // non-null: the run ends a connected component, its single exit value goes here
	uint64_t pad;                    // host side: index of the run in the plan
	const uint32_t* tab;             // DevProblem::slot_tab / pslot_tab of the owning table
	const void* rows;                // DevProblem::slot_rows / pslot_rows
	const uint32_t* ctrl;            // DevProblem::slot_ctrl
	uint8_t* bt;                     // the table's backtrace arena
	unsigned long long* spec_keys;   // DevProblem::spec_keys
	uint32_t spec_stride, pad2;      // pad2: WHAMD_SLOT_SKIP flags of a timing experiment (group launches)
};
static_assert(sizeof(SlotBatchEntry) == 320 && sizeof(SlotBatchEntry) % 64 == 0, "entries are fetched with wide scalar loads, whole 64-byte lines");

// Kernel argument of a group launch: blockIdx.y selects the entry (a pointer into the owning table's own entry array -- the arrays are
// built once per table at create time; a group launch only passes which of them take part).
constexpr int SLOT_GROUP_MAX = 248;   // (2 KB of kernel arguments: a whole-genome cohort is hundreds of tables)
struct SlotGroupArgs {
	uint32_t n, pad;   // pad != 0: the TABLE is the fast grid dimension (blockIdx.x, padded to a multiple of eight): a table's workgroups on one XCD (slot_group_who)
	const SlotBatchEntry* entry[SLOT_GROUP_MAX];
};

// Backtrace side of a run's column: which slot holds the read of every logical bit, and how many ending reads of the
// run lie in earlier columns (the path's state at this column = the state after undoing every later ending read).
struct SlotBtCol {
	uint8_t k, kf, pad[2];
	uint8_t slot[28];                // [k] slot of the read at logical position j
};
static_assert(sizeof(SlotBtCol) == 32, "SlotBtCol must stay 8 dwords");

// Header of a slot run in the backtrace unit list (same 128 bytes as BtUnit, kind == 2).
struct SlotBtUnit {
	uint32_t kind, c0, ncols, blob_off;    // blob_off: word offset of the run's backtrace blob ([ncols] SlotBtCol, then the local
	                                       // slot of every ending read in forward order, one byte each, padded to a word)
	uint32_t g, L, n_ends, threads;
	uint32_t bt_lo, bt_hi, half, blob_words;
	uint32_t f_exit, lr, pad0[2];
	uint8_t exit_slot[32];                 // [f_exit] slot of the read at bit j of the logical exit index
	uint8_t exit_pos[32];                  // [f_exit] bit of the exit (exchange) index that holds bit j of the logical exit index
};
static_assert(sizeof(SlotBtUnit) == 128, "SlotBtUnit must stay 32 words");

// (rows and backtrace columns: vectors that are sized without being written, on huge pages -- host_parallel.h)
template <class T>
using NoInitAllocator = NoInitAlloc<T>;

struct SlotPlan {
	std::vector<Step> steps;                 // kind 0: per-column step (index = column), kind 2: slot run (index into runs)
	std::vector<SlotRun> runs;
	std::vector<SlotRow, NoInitAllocator<SlotRow>> rows;         // indexed by column (rows of columns outside runs are unused)
	std::vector<SlotBtCol, NoInitAllocator<SlotBtCol>> bt_cols;  // parallel to rows
	std::vector<uint8_t> end_slots;          // per run (end_off): local slot of every ending read, forward order
	std::vector<uint32_t> end_off;           // per run: first byte in end_slots
	std::vector<uint32_t> f_exit;            // per run: bits of the logical exit index
	std::vector<std::vector<uint8_t>> exit_slot;  // per run: slot of the read at bit j of the logical exit index
	std::vector<uint32_t> ctrl;              // per run SLOT_CTRL_WORDS words (SlotRun::ctrl_off): one byte per column, n_end | slot of
	                                         // the first ending read << 2 -- what steers the kernel's control flow, kept in SGPRs
	std::vector<int32_t> col_to_row;         // [n_cols] index into rows or -1
	std::vector<uint32_t> component_first_step;
	uint64_t n_run_columns = 0;
	// pedigree tables (T > 1): rows of the run columns (indexed by column, like `rows`), per-run extras, words of all tables
	bool ped = false;
	// genotype_mode: per run (start_off) the local slot of every read that STARTS inside the run, forward order; per column
	// PedSlotRow::pad[0] = reads starting in the column, pad[1] = how many started in earlier columns of the run
	std::vector<uint8_t> start_slots;
	std::vector<uint32_t> start_off;
	std::vector<PedSlotRow, NoInitAllocator<PedSlotRow>> prows;
	std::vector<PedSlotExtra> pextra;
	uint64_t table_words = 0;
};

// Plans the forward pass of a single-individual table with slot runs wherever they apply (per-column steps elsewhere).
// Returns false if the table is not eligible (values beyond 2^30; a pedigree other than one or two trios, or one whose
// columns mostly need more than PSLOT_MAXFORMS forms per transmission value): the caller uses plan_forward().
// use_symmetry: 0 never halve, >= 1 halve every run whose columns are symmetric and that has a grid slot.
// A table with T = 4 or 16 gets pedigree slot runs (plan.ped; l_pref and lr are then ignored unless l_pref < 0: -l_pref
// local slots, for tests).
// genotype_mode (genotype_slots.hip): the same one-value-per-lane layout for the forward-backward genotyper, for any T in {1, 4, 16}
// (T = 1: six lane slots); only the slot assignment, the ending reads (prows / ctrl / bt_cols) and the exchange layouts are
// planned -- no cost forms (Problem::terms is not needed), no tables.
bool plan_forward_slots(const Problem& p, int l_pref, int use_symmetry, SlotPlan& plan, int lr = 2, bool genotype_mode = false);

// Host-only diagnostic (slot_emulate.cpp): executes `plan` cell by cell the way the kernels do.  For planner tests on
// small inputs; never part of a solve.
bool emulate_slot_plan(const Problem& p, const SlotPlan& plan, std::vector<uint32_t>& path_index, uint32_t& score, std::string& msg);
// The same for a pedigree plan (pedslot_emulate.cpp): tables, butterfly min-plus, byte records, walk -- as kernels_pedslots.h does it.
bool emulate_pedslot_plan(const Problem& p, const SlotPlan& plan, std::vector<uint32_t>& path_index, std::vector<uint32_t>& path_trans,
                          uint32_t& score, std::string& msg);
// Host restatement of one table entry (pedslot_tables computes the same on the device): kind 0 G, 1 W, 2 S.
uint32_t pedslot_table_entry(const Problem& p, const SlotPlan& plan, uint32_t run_index, int kind, uint32_t unit, uint32_t c, uint32_t t, uint32_t f);

}  // namespace whamd
