// device_types.h -- plain structs shared between the host driver and the gfx950 kernels.
#pragma once
#include <cstdint>

#include "resident.h"
#include "slots.h"

namespace whamd {

// Packed formats shared by the per-column kernels and the backtrace:
//   key of a projection entry  = value << 32 | gray_rank(x) << KEY_JBITS | argj      (rank < 2^26: at most 25 reads per column)
//   raw backtrace record (u32) = the low word of the key
//   state of the walk (u32)    = logical index | transmission value << BT_STATE_TSHIFT
constexpr uint32_t KEY_JBITS = 6, KEY_JMASK = 63u;          // argj of up to 64 transmission values (three trios)
constexpr uint32_t BT_STATE_TSHIFT = 26, BT_STATE_XMASK = 0x03FFFFFFu;

// Per-column descriptor, read wave-uniformly (scalar loads) by every kernel.
struct DevColumn {
	uint32_t k;          // active reads (bits of a bipartition index)
	uint32_t b;          // backward width: low b bits index the previous projection
	uint32_t f;          // forward width: bits that survive into the next column (0 for the last column)
	uint32_t recomb;     // recombcost[c]
	uint32_t delta_off;  // into DevProblem::delta: [n_ind][k] signed per-bit deltas
	uint32_t term_off;   // into DevProblem::term_ptr: T+1 offsets into DevProblem::terms
	uint32_t seg_off;    // into DevProblem::segs: nseg_fwd forward segments, then nseg_end ending segments
	uint16_t nseg_fwd, nseg_end;
	uint32_t mode;       // 0: fused column step, bit-plane backtrace; 1: key (atomic) path, raw u32 backtrace;
	                     // 2: resident run (resident.h), per-workgroup bit planes; 3: slot run (slots.h)
	uint32_t ebits;      // k - f: reads that end in this column
	uint32_t eloop;      // log2 of the ending-bit patterns each thread enumerates itself
	uint32_t nplanes;    // mode 0: ebits + transmission bits
	uint64_t bt_off;     // byte offset of this column's backtrace record in the arena
	uint32_t is_last;
	uint32_t res_idx;    // mode 2: index into DevProblem::res_bt / res_cols
};

struct DevTerm {
	uint32_t c, plus, minus;
};

struct DevProblem {
	const DevColumn* cols;
	const int32_t* delta;
	const uint32_t* term_ptr;
	const DevTerm* terms;
	const uint32_t* segs;   // packed: src_shift | dst_shift << 8 | len << 16
	uint8_t* bt;            // backtrace arena
	unsigned long long* keys;       // [2^max_f * T] scratch of the key path, all-ones between uses
	unsigned long long* last_keys;  // [T] keys of the last column
	// resident path (resident.h)
	const ResColumn* res_cols;
	const ResBacktrace* res_bt;
	const PedColumn* ped_cols;   // trio runs: descriptors parallel to res_cols
	const PedTerm* ped_terms;    // trio runs: term pool
	int32_t* ped_tables;    // [trio columns][PED_TABLE] lookup tables (computed at the start of each solve)
	int32_t* res_tables;    // [resident columns][RES_TABLE] lookup tables (computed at the start of each solve)
	// slot runs (slots.h)
	const SlotRow* slot_rows;    // per-column descriptors of the slot runs
	const uint32_t* slot_blob;   // backtrace blobs of the slot runs (SlotBtUnit::blob_off)
	const uint32_t* slot_ctrl;   // control bytes of the slot runs (SlotRun::ctrl_off)
	const uint32_t* slot_tab;    // per-run tables of the single-individual slot runs (SlotRun::tab_g / tab_w / tab_sl)
	const PedSlotRow* pslot_rows;   // pedigree slot runs: per-column descriptors (indexed by column)
	const uint32_t* pslot_tab;      // pedigree slot runs: the cost-form tables G / W / S of every run (PedSlotExtra offsets)
	unsigned long long* spec_keys;  // [chunk boundary][spec_stride]: per wave of the boundary run, min over the cells it stored of
	                                // (value << 32 | exit index); all-ones before
	uint32_t spec_stride;
	unsigned long long* dbg;  // optional cycle-counter dump (WHAMD_DEBUG_STAMPS / WHAMD_SLOT_STAMPS)
	uint32_t dbg_wg_off;      // word offset of the per-workgroup start/end stamps inside dbg
	uint32_t dbg_flags;       // experiments: bit 0 skip the slice store, bit 1 skip the record store (results invalid)
	uint32_t n_cols;
	uint32_t T;
	uint32_t tbits;         // 2 * triples
	uint32_t n_ind;
	uint32_t* bt_state;     // windowed solve (DeviceTable::Impl::Window): (x, transmission value) at the oldest column walked so far
};

}  // namespace whamd
