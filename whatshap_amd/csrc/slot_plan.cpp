// slot_plan.cpp -- host planner of the register-resident forward path (slots.h): cuts the column chain of a
// single-individual table into runs, assigns every read of a run its slot, builds the per-column descriptors, the
// exchange layouts between steps and everything the backtrace needs.
//
// Reference semantics restated here: ColumnIndexingScheme (src/columnindexingscheme.cpp:7-34,62-85: the reads shared with
// the previous column are the low b bits, the forward mask compacts the reads that continue), the cost terms of
// PedigreeColumnCostComputer (src/pedigreecolumncostcomputer.cpp:14-114, closed form of DESIGN.md section 2) and the
// Gray-rank tie rule of src/pedigreedptable.cpp:306-327.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <iterator>
#include <cstdio>
#include <thread>

#include "debug_build.h"
#include "slots.h"

namespace whamd {

namespace {

struct RunDraft {
	uint32_t c0 = 0, ncols = 0, g = 0, L = 0, lw = 0;
	std::vector<int32_t> entry_read;   // [L + g] read id per slot at entry (-1 free)
	std::vector<int32_t> exit_read;    // [L + g] read id per slot after the last column
	bool symmetric = true;
};

inline uint32_t parity32(uint32_t v) { return (uint32_t)__builtin_popcount(v) & 1u; }

}  // namespace

bool plan_forward_slots(const Problem& p, int l_pref, int use_symmetry, SlotPlan& plan, int lr, bool genotype_mode) {
	const auto tp0 = std::chrono::steady_clock::now();
	const long pf0 = thread_minor_faults();
	plan = SlotPlan();
	const uint32_t n = p.n_cols;
	// pedigree tables (one or two trios): a lane holds ONE (cell, transmission value); no reg slots, 6 - TB lane slots
	const bool ped = p.T > 1 || genotype_mode;   // (one value per lane; genotype_mode also for a single individual)
	const uint32_t TB = p.T == 4 ? 2u : (p.T == 16 ? 4u : 0u);
	if (ped && !genotype_mode && (TB == 0 || p.n_ind < 3)) return false;
	if (genotype_mode && p.T != 1 && TB == 0) return false;
	if (!ped && !(p.T == 1 && p.n_ind == 1)) return false;
	if (!genotype_mode && !(p.value_bound < 1073741824.0)) return false;
	plan.ped = ped;
	// a trio / a quartet with untrusted genotypes: factorised lines (PSLOT_FACT, PSLOT_FACT4)
	const uint32_t fact_nf = (ped && !genotype_mode && !p.fterms.empty()) ? (TB == 2 && p.fterm_kind == 1 ? (uint32_t)PSLOT_FACT : (TB == 4 && p.fterm_kind == 2 ? (uint32_t)PSLOT_FACT4 : 0u)) : 0u;
	const bool fact = fact_nf != 0;
	lr = ped ? 0 : std::max(1, std::min(lr, SLOT_LR));
	const int n_lane = ped ? 6 - (int)TB : SLOT_LANE;
	plan.col_to_row.assign(n, -1);
	const int LMIN = lr + n_lane, LMAX = lr + n_lane + SLOT_LWMAX;
	if (ped) l_pref = l_pref < 0 ? -l_pref : LMAX;
	l_pref = std::max(LMIN, std::min(l_pref, LMAX));
	const uint32_t max_run_cols = ped ? (uint32_t)PSLOT_MAXCOLS : (uint32_t)SLOT_MAXCOLS;
	const std::vector<uint32_t>& last_col = p.read_last_col;   // (problem.cpp)
	std::vector<int32_t>& col_to_row = plan.col_to_row;
	// rows and backtrace columns are indexed by COLUMN (row of column c = rows[c]; entries of columns outside runs stay unused):
	// the ranges below write disjoint parts of them in place
	if (ped) plan.prows.resize(n);
	else {
		plan.rows.reserve((size_t)n + SLOT_ROW_PAD);   // (the driver appends the pad rows: no reallocation of 50 MB)
		plan.rows.resize(n);
	}
	plan.bt_cols.resize(n);
	auto& rows_g = plan.rows;
	auto& prows_g = plan.prows;
	auto& btc_g = plan.bt_cols;
	// A run may start at ANY column (it reads the exchange column its predecessor left), so disjoint column ranges are
	// planned independently -- in parallel -- and concatenated; a range boundary is just one more run boundary.
	auto plan_range = [&](const uint32_t c_begin, const uint32_t c_end, SlotPlan& plan, std::vector<RunDraft>& drafts) {
	std::vector<int8_t> slot_of(p.n_reads, -1);
	SlotRow scratch_row{};       // (the row kind this table does not use is written here)
	PedSlotRow scratch_prow{};
	uint32_t c = c_begin;
	while (c < c_end) {
		auto column_step = [&]() { plan.steps.push_back(Step{0, c}); ++c; };
		if (c + 1 >= n && !genotype_mode) { column_step(); continue; }   // the last column needs the global optimum (column_step_keys)
		const uint32_t b0 = p.b[c];
		const ColumnEntry* first = p.col_begin(c);
		// ---- shape of the run: local slots L, grid slots g.  A grid read must outlive the run, a starting read needs a free local slot:
		// few grid slots end the run when the coverage grows, many end it when the first of them does.  The widest column of the next
		// 1 .. 12 columns gives the candidates for g; each is probed (slot counts only) and the one with the longest run is planned.
		constexpr uint32_t LOOK = 12;
		uint32_t kmax_at[LOOK], n_look = 0;
		for (uint32_t cc = c; cc < std::min(n, c + LOOK); ++cc) {
			if (cc > c && p.b[cc] == 0) break;
			kmax_at[n_look] = std::max<uint32_t>(n_look ? kmax_at[n_look - 1] : 0u, p.k[cc]);
			++n_look;
		}
		const uint32_t kmax = kmax_at[std::min<uint32_t>(n_look, genotype_mode ? 6u : LOOK) - 1];
		const uint32_t L = std::min<uint32_t>((uint32_t)l_pref, std::max<uint32_t>((uint32_t)LMIN, kmax));
		uint32_t g = kmax > L ? kmax - L : 0;
		if (genotype_mode) g = std::min(g, b0);   // (a short run while the coverage ramps up rather than a column the run kernels cannot take)
		// the entering reads by the column they end in, latest first (ties: the younger read): the first g of them are the grid reads
		std::vector<uint32_t> order(b0);
		for (uint32_t j = 0; j < b0; ++j) order[j] = j;
		std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t bb) {
			const uint32_t ea = last_col[first[a].read_id], eb = last_col[first[bb].read_id];
			if (ea != eb) return ea > eb;
			return a > bb;
		});
		if (!genotype_mode) {
			auto probe = [&](uint32_t gg) -> uint32_t {   // columns a run with gg grid slots would take (the walk below, counts only)
				uint32_t grid_end = 0xFFFFFFFFu;
				for (uint32_t i = 0; i < gg; ++i) grid_end = std::min(grid_end, last_col[first[order[i]].read_id]);
				uint32_t n_free = L - (b0 - gg), ends = 0, c1 = c;
				for (; c1 < n && c1 - c < max_run_cols; ++c1) {
					if (c1 + 1 == n || c1 >= grid_end || c1 >= c_end || (c1 > c && p.b[c1] == 0)) break;
					const uint32_t kc = p.k[c1], bc = c1 == c ? b0 : p.b[c1], n_new = kc - bc, n_end = kc - p.f[c1];
					if (n_end > (uint32_t)(ped ? PSLOT_MAXEND : SLOT_MAXEND) || ends + n_end > (uint32_t)SLOT_MAXENDS_RUN || n_new > n_free) break;
					n_free = n_free - n_new + n_end;
					ends += n_end;
				}
				return c1 - c;
			};
			uint32_t best_g = 0xFFFFFFFFu, best_len = 0, last_g = 0xFFFFFFFFu;
			for (uint32_t q = 0; q < n_look; ++q) {
				const uint32_t gg = kmax_at[q] > L ? kmax_at[q] - L : 0;
				if (gg == last_g) continue;
				last_g = gg;
				if (gg > b0 || gg > (uint32_t)SLOT_GMAX || p.k[c] > L + gg || b0 - gg > L) continue;
				const uint32_t len = probe(gg);
				if (len > best_len) { best_len = len; best_g = gg; }
			}
			if (best_g == 0xFFFFFFFFu) { column_step(); continue; }
			g = best_g;
		}
		if (g > b0 || g > (uint32_t)SLOT_GMAX || p.k[c] > L + g) { column_step(); continue; }
		if (genotype_mode) {
			// g was sized for the columns ahead; a grid read must outlive this column, or the run would be empty (every column has
			// to lie in a run here): fewer grid slots if the reads allow it
			while (g > 0 && c + 1 < n && last_col[first[order[g - 1]].read_id] <= c) --g;
			if (p.k[c] > L + g) { column_step(); continue; }
		}
		RunDraft d;
		d.c0 = c; d.g = g; d.L = L; d.lw = L - (uint32_t)LMIN;
		const uint32_t nslots = L + g;
		d.entry_read.assign(nslots, -1);
		uint32_t grid_end = 0xFFFFFFFFu;
		{
			std::vector<uint32_t> grid_reads;
			for (uint32_t i = 0; i < g; ++i) {
				grid_reads.push_back(first[order[i]].read_id);
				grid_end = std::min(grid_end, last_col[first[order[i]].read_id]);
			}
			// Grid slots (bit i of the workgroup index).  Workgroup b runs on XCD b % 8 (observed placement; speed only) and
			// each XCD has its own L2, so the reads that the NEXT run will keep in its lowest slots -- the ones that end
			// first -- must not sit in the three low bits: workgroups that share a 128-byte line of the exit column then share
			// an L2, which merges their 16-byte pieces into whole lines before the write-back.  Slots 0..2 and the top slot
			// (the halved one) therefore take the reads that end LAST; the others follow in ascending end order.
			std::stable_sort(grid_reads.begin(), grid_reads.end(), [&](uint32_t a, uint32_t bb) {
				if (last_col[a] != last_col[bb]) return last_col[a] < last_col[bb];
				return a < bb;
			});
			if (g >= 5) {
				d.entry_read[L + g - 1] = (int32_t)grid_reads[g - 1];
				for (uint32_t i = 0; i < 3; ++i) d.entry_read[L + i] = (int32_t)grid_reads[g - 4 + i];
				for (uint32_t i = 3; i + 1 < g; ++i) d.entry_read[L + i] = (int32_t)grid_reads[i - 3];
			} else {
				for (uint32_t i = 0; i < g; ++i) d.entry_read[L + i] = (int32_t)grid_reads[i];
			}
			// local entering reads: the one that ends first gets the lowest slot (reg slots first, wave slots last)
			uint32_t s = 0;
			for (uint32_t i = b0; i-- > g;) d.entry_read[s++] = (int32_t)first[order[i]].read_id;
		}
		std::vector<int32_t> cur = d.entry_read;   // read per slot, updated column by column
		auto release_marks = [&]() {
			for (int32_t r : d.entry_read) if (r >= 0) slot_of[r] = -1;
			for (int32_t r : cur) if (r >= 0) slot_of[r] = -1;
		};
		for (uint32_t s = 0; s < nslots; ++s) if (cur[s] >= 0) slot_of[cur[s]] = (int8_t)s;
		// ---- walk the columns
		const size_t rows_mark = c, ends_mark = plan.end_slots.size(), starts_mark = plan.start_slots.size();
		std::vector<int32_t> started;   // reads that got their slot inside the run (marks to clear)
		uint32_t c1 = c, n_ends = 0;
		bool symmetric = true;
		uint32_t run_forms = 2;   // pedigree runs: NF = 2 or 4 forms per transmission value
		while (c1 < n && c1 - c < max_run_cols) {
			if (c1 + 1 == n && !genotype_mode) break;
			if (c1 >= grid_end && !(genotype_mode && c1 + 1 == n)) break;   // (genotyping: nothing follows the last column, its reads need not be summed out)
			if (c1 >= c_end) break;
			if (c1 > c && p.b[c1] == 0) break;
			const ColumnEntry* col = p.col_begin(c1);
			const uint32_t kc = p.k[c1], bc = c1 == c ? b0 : p.b[c1];
			const uint32_t n_new = kc - bc, n_end = kc - p.f[c1];
			if (!genotype_mode && (n_end > (uint32_t)(ped ? PSLOT_MAXEND : SLOT_MAXEND) || n_ends + n_end > (uint32_t)SLOT_MAXENDS_RUN)) break;
			if (genotype_mode && n_ends + n_end > 250u) break;
			uint32_t n_free = 0;
			for (uint32_t s = 0; s < L; ++s) n_free += cur[s] < 0;
			if (n_new > n_free) break;
			bool ok = true;
			if (!ped) {   // (one pass, no early exit: every delta within the limit and every shared read tracked)
				const int32_t* dcol = p.delta.data() + (size_t)p.col_ptr[c1];
				uint32_t bad = 0;
				for (uint32_t j = 0; j < kc; ++j) bad |= (uint32_t)(std::abs(dcol[j]) >= SLOT_DELTA_LIMIT);
				for (uint32_t j = 0; j < bc; ++j) bad |= (uint32_t)(slot_of[col[j].read_id] < 0);
				ok = bad == 0;
			}
			if (ped && !genotype_mode) {
				// at most PSLOT_MAXFORMS forms per transmission value; the run's tables grow with the widest column (NF 2 -> 4)
				uint32_t most = 0;
				for (uint32_t t = 0; t < p.T; ++t) most = std::max<uint32_t>(most, (uint32_t)(p.term_end(c1, t) - p.term_begin(c1, t)));
				if (!fact && (most > (uint32_t)PSLOT_MAXFORMS || (most > 4u && TB != 2u))) {   // (sixteen forms: the trio kernel only)
					if (debug_env("WHAMD_DEBUG_PLAN")) fprintf(stderr, "[plan] column %u: %u cost forms per transmission value: no pedigree run\n", c1, most);
					break;
				}
				const uint32_t nf = fact ? pslot_na(fact_nf) : std::max(run_forms, most > 4 ? 16u : (most > 2 ? 4u : 2u));
				if ((c1 - c + 1) * pslot_ta(fact_nf, p.T) * nf > (uint32_t)PSLOT_FORMWORDS) break;
				run_forms = nf;
			}
			if (ped) for (uint32_t j = 0; j < bc && ok; ++j) ok = slot_of[col[j].read_id] >= 0;   // every shared read is tracked
			if (!ok) break;
			// reads that start here take the lowest free local slots
			{
				uint32_t s = 0;
				for (uint32_t j = bc; j < kc; ++j) {
					while (cur[s] >= 0) ++s;
					cur[s] = (int32_t)col[j].read_id;
					slot_of[col[j].read_id] = (int8_t)s;
					started.push_back((int32_t)col[j].read_id);
					if (genotype_mode) plan.start_slots.push_back((uint8_t)s);
				}
			}
			const int32_t* dl = genotype_mode ? nullptr : p.delta.data() + (size_t)p.col_ptr[c1] * p.n_ind;   // [individual][bit]; n_ind == 1 unless ped
			// the row is built IN PLACE (a row of a column that ends up outside every run is never read); the other kind goes to a scratch row
			SlotRow& row = ped ? scratch_row : rows_g[c1];
			PedSlotRow& prow = ped ? prows_g[c1] : scratch_prow;
			if (ped) prow = PedSlotRow{}; else row = SlotRow{};
			SlotBtCol bc_rec{};
			bc_rec.k = (uint8_t)kc;
			bc_rec.kf = (uint8_t)n_ends;
			if (ped) {
				// the cost forms stay where build_problem left them (DevProblem::terms); the row only says which read sits where
				prow.recomb = p.recomb[c1];
				for (uint32_t j = 0; j < kc; ++j) {
					const int s = slot_of[col[j].read_id];
					if (!genotype_mode) prow.dslot[s] = dl[(size_t)col[j].sample * kc + j];
					prow.ind[s] = col[j].sample;
					bc_rec.slot[j] = (uint8_t)s;
				}
				symmetric = false;
			} else {
				uint32_t Cp = RES_ABSENT, Cm = RES_ABSENT, Cc = INF;
				for (uint64_t q = p.term_begin(c1, 0); q < p.term_end(c1, 0); ++q) {
					const CostTerm& t = p.terms[q];
					if (t.plus) Cp = t.c;
					else if (t.minus) Cm = t.c;
					else Cc = std::min(Cc, t.c);
				}
				row.K = Cp + Cm;
				row.Cc = Cc;
				row.Cp = Cp;
				uint32_t dsum = 0;
				for (uint32_t j = 0; j < kc; ++j) {
					const int s = slot_of[col[j].read_id];
					row.dslot[s] = dl[j];
					dsum += (uint32_t)dl[j];
					bc_rec.slot[j] = (uint8_t)s;
				}
				for (int s = 0; s < lr; ++s) row.dreg[s] = row.dslot[s];
				for (int s = 0; s < SLOT_LANE; ++s) row.dlane[s] = row.dslot[lr + s];
				// cost(~x) == cost(x)  <=>  Cp + (sum of all deltas) == Cm, or no orientation term at all
				const bool both_absent = Cp == RES_ABSENT && Cm == RES_ABSENT;
				if (!both_absent && (Cp == RES_ABSENT || Cm == RES_ABSENT || Cp + dsum != Cm)) symmetric = false;
			}
			// ending reads, ascending logical position
			uint32_t en = 0;
			const uint32_t ending_bits = ~p.fwd_mask[c1] & (kc >= 32 ? 0xFFFFFFFFu : ((1u << kc) - 1u));   // (one or two of fifteen: the set bits, not a scan)
			for (uint32_t rest = ending_bits; rest; rest &= rest - 1u) {
				const uint32_t j = (uint32_t)__builtin_ctz(rest);
				if (genotype_mode && c1 + 1 == n) break;   // (the last column of the table: nothing is summed out)
				const int s = slot_of[col[j].read_id];
				if (s >= (int)L) { ok = false; break; }   // a grid read would end (excluded by grid_end)
				uint32_t M = 0;
				for (uint32_t q = j + 1; q < kc; ++q) M |= 1u << slot_of[col[q].read_id];
				const uint32_t mflip = parity32(M);
				uint32_t qmask = 0;
				for (uint32_t r = 0; r < (1u << lr); ++r) {
					uint32_t bit = parity32(r & M);
					if (s < lr) bit ^= ((r >> s) & 1u) & mflip;
					qmask |= bit << r;
				}
				// (bit 0 of a qmask is never set: cell 0 of a thread has no reg-slot bit -- the X runs' ending block relies on it, kernels_slots.h)
				if (qmask & 1u) { ok = false; break; }
				if (en < (uint32_t)SLOT_MAXEND) {
					row.end[en].info = (uint32_t)s | (qmask << 8) | (mflip << 24);
					// (a read in a lane / wave slot: the kernel's parity also takes (side & mflip) -- the side is bit s of the thread's index,
					// which is not in M, so it is folded into the mask: one AND + popcount instead of a shift, two ANDs, a select and a XOR)
					row.end[en].M = s >= lr ? M ^ (mflip << s) : M;
				}
				if (ped && !genotype_mode) {   // (the kernel's cell index has no reg bits: M as it is)
					if (en == 0) { prow.info0 = (uint32_t)s; prow.M0 = M; }
					else if (en == 1) { prow.info1 = (uint32_t)s; prow.M1 = M; }
					else if (en == 2) { prow.info2 = (uint32_t)s; prow.M2 = M; }
					else { prow.pad[2] = (uint32_t)s; prow.pad[3] = M; }
					// the backtrace walks column by column: ending slots of the column, in order (the fourth next to the count)
					if (en < 3) bc_rec.slot[25 + en] = (uint8_t)s; else bc_rec.pad[1] = (uint8_t)s;
				}
				plan.end_slots.push_back((uint8_t)s);
				++en;
			}
			if (!ok) break;
			row.n_end = en;
			prow.n_end = en;
			if (ped) bc_rec.pad[0] = (uint8_t)en;
			if (genotype_mode) {
				prow.pad[0] = n_new;
				prow.pad[1] = (uint32_t)(plan.start_slots.size() - starts_mark) - n_new;
			}
			// after the projection the ended reads' slots are free again
			for (uint32_t rest = ending_bits; rest; rest &= rest - 1u) cur[slot_of[col[__builtin_ctz(rest)].read_id]] = -1;
			n_ends += en;
			col_to_row[c1] = (int32_t)c1;
			btc_g[c1] = bc_rec;
			++c1;
		}
		// a run whose bookkeeping stopped in the middle of a column: drop what that column appended
		{
			size_t keep = 0;
			for (size_t i = rows_mark; i < c1; ++i) keep += ped ? prows_g[i].n_end : rows_g[i].n_end;
			plan.end_slots.resize(ends_mark + keep);
			if (genotype_mode) {
				size_t keep_starts = 0;
				for (size_t i = rows_mark; i < c1; ++i) keep_starts += prows_g[i].pad[0];
				plan.start_slots.resize(starts_mark + keep_starts);
			}
		}
		if (c1 - c < (genotype_mode ? 1u : 2u)) {   // not worth a launch of its own
			for (uint32_t cc = c; cc < c1; ++cc) col_to_row[cc] = -1;
			plan.end_slots.resize(ends_mark);
			plan.start_slots.resize(starts_mark);
			release_marks();
			for (int32_t r : started) slot_of[r] = -1;
			column_step();
			continue;
		}
		// exit state: the reads that continue after column c1 - 1 (recomputed: `cur` may hold marks of a dropped column)
		d.exit_read.assign(nslots, -1);
		{
			const ColumnEntry* col = p.col_begin(c1 - 1);
			for (uint32_t j = 0; j < p.k[c1 - 1]; ++j)
				if ((p.fwd_mask[c1 - 1] >> j) & 1u) d.exit_read[btc_g[c1 - 1].slot[j]] = (int32_t)col[j].read_id;
		}
		release_marks();
		for (int32_t r : started) slot_of[r] = -1;
		d.ncols = c1 - c;
		d.symmetric = symmetric;
		SlotRun run{};
		run.c0 = c; run.ncols = d.ncols; run.g = g; run.L = L; run.lw = d.lw;
		run.ctrl_off = (uint32_t)plan.ctrl.size();
		plan.ctrl.resize(plan.ctrl.size() + SLOT_CTRL_WORDS, 0);
		uint32_t exchanges = 0;   // wave-slot endings so far (single individual): they alternate between two LDS exchange buffers -- which one travels with the ending read
		for (uint32_t i = 0; i < d.ncols; ++i) {
			if (!ped) {
				SlotRow& xrow = rows_g[rows_mark + i];
				for (uint32_t k = 0; k < std::min<uint32_t>(xrow.n_end, SLOT_MAXEND); ++k) {
					if ((xrow.end[k].info & 31u) < (uint32_t)(lr + SLOT_LANE)) continue;
					if (exchanges & 1u) xrow.end[k].info |= 1u << 16;   // (bits 16 .. 23 of info are free: slot | qmask << 8 | buffer << 16 | mflip << 24)
					++exchanges;
				}
			}
			const uint32_t ne = ped ? prows_g[rows_mark + i].n_end : rows_g[rows_mark + i].n_end;
			const uint32_t s0 = ped ? prows_g[rows_mark + i].info0 : (rows_g[rows_mark + i].end[0].info & 31u);
			const uint32_t byte = std::min(ne, 3u) | ((ne ? (s0 & 31u) : 0u) << 2);   // (3: three or more, the kernel reads the row's count)
			if (ped) plan.ctrl[run.ctrl_off + (i >> 2)] |= byte << ((i & 3u) * 8u);
			else {   // single individual: 16 bits per column, the first ending read's qmask (tie parity of the thread's cells) next to its slot
				const uint32_t qm = ne ? ((rows_g[rows_mark + i].end[0].info >> 8) & 255u) : 0u;
				const uint32_t xbuf = ne ? ((rows_g[rows_mark + i].end[0].info >> 16) & 1u) : 0u;   // (bit 15: read by the X runs only, kernels_slots.h)
				plan.ctrl[run.ctrl_off + (i >> 1)] |= (byte | (qm << 7) | (xbuf << 15)) << ((i & 1u) * 16u);
			}
		}
		run.lr = (uint32_t)lr;
		run.row_off = (uint32_t)rows_mark;
		run.n_ends = 0;
		for (size_t i = rows_mark; i < c1; ++i) run.n_ends += ped ? prows_g[i].n_end : rows_g[i].n_end;
		if (ped) {
			PedSlotExtra ex{};
			ex.tb = TB;
			ex.nf = 2;   // (recomputed: run_forms may have grown for a column that was dropped again)
			for (uint32_t i = 0; i < d.ncols && !genotype_mode; ++i)
				for (uint32_t t = 0; t < p.T; ++t) {
					const uint32_t cnt = (uint32_t)(p.term_end(c + i, t) - p.term_begin(c + i, t));
					ex.nf = std::max(ex.nf, cnt > 4 ? 16u : (cnt > 2 ? 4u : 2u));
				}
			if (fact) ex.nf = fact_nf;
			if (fact_nf == (uint32_t)PSLOT_FACT4) { ex.pad[0] = p.fact4_roles[0]; ex.pad[1] = p.fact4_roles[1]; }
			ex.fwn = d.ncols * pslot_ta(ex.nf, p.T) * pslot_na(ex.nf);
			ex.arow = (ex.fwn + 3u) & ~3u;
			ex.rec_words = ((d.ncols + 3u) / 4u) * (64u << d.lw);
			plan.pextra.push_back(ex);
		}
		run.threads = 64u << d.lw;
		run.has_prev = c > 0;   // a run that starts a connected component reads the single value the previous one projected onto
		run.half = (use_symmetry > 0 && symmetric && g >= 1) ? 1u : 0u;
		for (uint32_t s = 0; s < nslots; ++s) {
			if (d.entry_read[s] >= 0) run.in_occ |= 1u << s;
			if (d.exit_read[s] >= 0) run.out_occ |= 1u << s;
		}
		plan.steps.push_back(Step{2, (uint32_t)plan.runs.size()});
		plan.runs.push_back(run);
		plan.end_off.push_back((uint32_t)ends_mark);
		plan.start_off.push_back((uint32_t)starts_mark);
		drafts.push_back(d);
		plan.n_run_columns += d.ncols;
		c = c1;
	}
	};
	std::vector<RunDraft> drafts;
	const auto tp1 = std::chrono::steady_clock::now();
	const long pf1 = thread_minor_faults();
	long pf2 = pf1;
	auto tp2 = tp1;
	{
		// The ranges planned independently become run boundaries: they follow from the INPUT alone (fixed pieces of ~PLAN_PIECE columns), not
		// from how many host threads this machine has -- the same table gets the same runs, launches and record layout everywhere.
		constexpr uint32_t PLAN_PIECE = 8192;
		const uint32_t n_pieces = std::max<uint32_t>(1, n / PLAN_PIECE);
		std::vector<SlotPlan> parts(n_pieces);
		std::vector<std::vector<RunDraft>> part_drafts(n_pieces);
		std::vector<uint32_t> bounds(n_pieces + 1);
		for (uint32_t t = 0; t <= n_pieces; ++t) bounds[t] = (uint32_t)((uint64_t)n * t / n_pieces);
		parallel_ranges(n_pieces, host_threads(n_pieces, 1), [&](uint64_t q0, uint64_t q1, uint32_t) {
			for (uint64_t q = q0; q < q1; ++q) plan_range(bounds[q], bounds[q + 1], parts[q], part_drafts[q]);
		});
		tp2 = std::chrono::steady_clock::now();
		pf2 = thread_minor_faults();
		const uint32_t n_threads = n_pieces;
		{
			size_t n_steps = 0, n_runs = 0, n_ends_all = 0, n_starts = 0, n_ctrl = 0, n_extra = 0;
			for (const SlotPlan& q : parts) {
				n_steps += q.steps.size(); n_runs += q.runs.size(); n_ends_all += q.end_slots.size(); n_starts += q.start_slots.size();
				n_ctrl += q.ctrl.size(); n_extra += q.pextra.size();
			}
			plan.steps.reserve(n_steps); plan.runs.reserve(n_runs); plan.end_off.reserve(n_runs); plan.start_off.reserve(n_runs);
			plan.end_slots.reserve(n_ends_all); plan.start_slots.reserve(n_starts); plan.ctrl.reserve(n_ctrl); plan.pextra.reserve(n_extra);
			drafts.reserve(n_runs);
		}
		for (uint32_t t = 0; t < n_threads; ++t) {
			const SlotPlan& q = parts[t];
			const uint32_t run_base = (uint32_t)plan.runs.size();
			const uint32_t end_base = (uint32_t)plan.end_slots.size(), ctrl_base = (uint32_t)plan.ctrl.size();
			for (Step st : q.steps) { if (st.kind == 2) st.index += run_base; plan.steps.push_back(st); }
			for (SlotRun run : q.runs) { run.ctrl_off += ctrl_base; plan.runs.push_back(run); }
			plan.pextra.insert(plan.pextra.end(), q.pextra.begin(), q.pextra.end());
			for (uint32_t off : q.end_off) plan.end_off.push_back(off + end_base);
			for (uint32_t off : q.start_off) plan.start_off.push_back(off + (uint32_t)plan.start_slots.size());
			plan.start_slots.insert(plan.start_slots.end(), q.start_slots.begin(), q.start_slots.end());
			plan.end_slots.insert(plan.end_slots.end(), q.end_slots.begin(), q.end_slots.end());
			plan.ctrl.insert(plan.ctrl.end(), q.ctrl.begin(), q.ctrl.end());
			plan.n_run_columns += q.n_run_columns;
			drafts.insert(drafts.end(), std::make_move_iterator(part_drafts[t].begin()), std::make_move_iterator(part_drafts[t].end()));   // (two small vectors per run: moved, not copied)
		}
	}
	const auto tp3 = std::chrono::steady_clock::now();
	const long pf3 = thread_minor_faults();
	for (size_t si = 0; si < plan.steps.size(); ++si) {
		const uint32_t c0 = plan.steps[si].kind == 2 ? plan.runs[plan.steps[si].index].c0 : plan.steps[si].index;
		// (a pedigree table is ONE job: across a column no read spans the T transmission values still couple the two sides)
		if (!ped && (si == 0 || p.b[c0] == 0)) plan.component_first_step.push_back((uint32_t)si);
	}
	// ---- Y form (kernels_slots.h).  A run whose columns all have both orientation terms and no constant one -- every column of a
	// `whatshap phase` table with trusted genotypes: only heterozygous variants are phased -- computes with Y = B_c - 2 D instead of D, where
	// B_c is the same for every cell of column c (the prefix sum below): min(A, K - A) = (K - |2A - K|) / 2, so a cell-column is ONE absolute
	// difference accumulated into Y (v_sad_u32) instead of subtract, min3, add; minima become maxima, the tie rule keeps its form.
	// Between two such runs the exchange column stays in Y form; at any other neighbour the run converts (D = (B - Y) / 2 exactly).
	if (!ped && (lr == 2 || lr == 3) && !debug_env("WHAMD_NO_YFORM")) {
		std::vector<uint8_t> pure(n, 0);
		std::vector<uint64_t> B((size_t)n + 1, 0);   // B[c + 1] = base after column c
		// (per column in parallel -- this pass, serial, was 4 of the 8.6 ms of planning configs[2] on 32 threads --, then one running sum)
		parallel_ranges(n, host_threads(n, 8192), [&](uint64_t c_lo, uint64_t c_hi, uint32_t) {
			for (uint32_t c = (uint32_t)c_lo; c < (uint32_t)c_hi; ++c) {
				uint64_t kstar = 0;
				uint32_t Cp = RES_ABSENT, Cm = RES_ABSENT, Cc = INF;
				for (uint64_t q = p.term_begin(c, 0); q < p.term_end(c, 0); ++q) {
					const CostTerm& t = p.terms[q];
					if (t.plus) Cp = t.c; else if (t.minus) Cm = t.c; else Cc = std::min(Cc, t.c);
				}
				pure[c] = Cp != RES_ABSENT && Cm != RES_ABSENT && Cc == INF;
				if (pure[c]) {
					kstar = (uint64_t)(uint32_t)(Cp + Cm);
				} else {
					// (the bound of a column that is not pure: its cheapest term at its largest -- the sum of the |deltas| is read only here, not for the
					//  heterozygous columns that make up a `whatshap phase` table)
					uint64_t dabs = 0, ub = ~0ull;
					const int32_t* dl = p.delta.data() + (size_t)p.col_ptr[c];
					for (uint32_t j = 0; j < p.k[c]; ++j) dabs += (uint64_t)std::abs((int64_t)dl[j]);
					for (uint64_t q = p.term_begin(c, 0); q < p.term_end(c, 0); ++q) {
						const CostTerm& t = p.terms[q];
						ub = std::min<uint64_t>(ub, (uint64_t)t.c + ((t.plus || t.minus) ? dabs : 0));
					}
					kstar = 2 * (ub == ~0ull ? 0 : ub);   // (2 D grows by at most this much in column c)
				}
				B[c + 1] = kstar;
			}
		});
		for (uint32_t c = 0; c < n; ++c) B[c + 1] += B[c];
		if (B[n] < 0xFFFFFFF0ull) {
			std::vector<uint8_t> yrun(plan.runs.size(), 0);
			parallel_ranges(plan.runs.size(), host_threads(plan.runs.size(), 512), [&](uint64_t r_lo, uint64_t r_hi, uint32_t) {
				for (size_t ri = r_lo; ri < r_hi; ++ri) {
					const SlotRun& run = plan.runs[ri];
					bool ok = run.lr == 2 || run.lr == 3;
					for (uint32_t i = 0; i < run.ncols && ok; ++i) ok = pure[run.c0 + i];
					yrun[ri] = ok;
				}
			});
			// (a step rewrites the rows of its own run only and reads its neighbours' flags: steps are independent)
			parallel_ranges(plan.steps.size(), host_threads(plan.steps.size(), 256), [&](uint64_t s_lo, uint64_t s_hi, uint32_t) {
			for (size_t si = s_lo; si < s_hi; ++si) {
				if (plan.steps[si].kind != 2 || !yrun[plan.steps[si].index]) continue;
				SlotRun& run = plan.runs[plan.steps[si].index];
				auto y_neighbour = [&](size_t sj) { return plan.steps[sj].kind == 2 && yrun[plan.steps[sj].index]; };
				// (a step that starts a connected component reads one value of the previous component, or starts from cost 0 as a job of its own)
				const bool in = si > 0 && p.b[run.c0] != 0 && y_neighbour(si - 1);
				bool out = false;
				if (si + 1 < plan.steps.size() && y_neighbour(si + 1)) out = p.b[plan.runs[plan.steps[si + 1].index].c0] != 0;
				run.yflags = 1u | (in ? 2u : 0u) | (out ? 4u : 0u);
				run.base_in = (uint32_t)B[run.c0];
				run.base_out = (uint32_t)B[run.c0 + run.ncols];
				for (uint32_t i = 0; i < run.ncols; ++i) {
					SlotRow& row = plan.rows[run.row_off + i];
					// Kr[r] = K - 2 * (sum of the reg-slot deltas set in r) + bias over the row's first 2^lr words (K, Cc, dreg[0..2], dlane[0..2]: the
					// run kernel reads none of them in Y form; slot_tables takes the lane deltas from dslot)
					const uint32_t K = row.K, dreg[3] = {(uint32_t)row.dreg[0], (uint32_t)row.dreg[1], (uint32_t)row.dreg[2]};
					uint32_t* words = reinterpret_cast<uint32_t*>(&row);
					for (uint32_t r = 0; r < (1u << run.lr); ++r) {
						uint32_t dsum = 0;
						for (uint32_t sb = 0; sb < run.lr; ++sb) if ((r >> sb) & 1u) dsum += dreg[sb];
						words[r] = K - 2u * dsum + SLOT_YBIAS;
					}
				}
			}
			});
		}
	}
	if (ped && !genotype_mode) {
		// a table that mostly falls back to per-column steps (genotypes not trusted: up to 16 forms per value) is better off
		// with the LDS-resident trio runs / the per-column kernels
		if (plan.n_run_columns * 2 < n) return false;
		// where each run's tables (G, W, S of slots.h) live in the table array
		uint64_t words = 0;
		for (size_t ri = 0; ri < plan.runs.size(); ++ri) {
			const SlotRun& run = plan.runs[ri];
			PedSlotExtra& ex = plan.pextra[ri];
			ex.g_lo = (uint32_t)words;
			ex.g_hi = (uint32_t)(words >> 32);
			// the min-plus step over the previous transmission value on packed keys value << TB | j (kernels_pedslots.h): every finite value of the table --
			// and 2 * triples * recomb of every column, part of the bound -- must stay below 2^(31 - TB) - 1
			if (p.value_bound < (double)((1u << (31u - TB)) - 1u) && !debug_env("WHAMD_NO_PED_KEYS")) plan.runs[ri].yflags |= 16u;
			const uint64_t gsz = ((uint64_t)1 << run.g) * ex.fwn;
			ex.w_off = (uint32_t)gsz;
			ex.s_off = (uint32_t)(gsz + ((uint64_t)1 << run.lw) * ex.fwn);
			words += (uint64_t)ex.s_off + (uint64_t)run.ncols * 64u * pslot_ns(ex.nf) + (uint64_t)run.ncols * p.T * pslot_nk(ex.nf);
			// X runs (kernels_pedslots.h, pedslot_runx): per-column scalars and the lanes' tie parities behind the cost tables.  An experiment of the debug
			// library only (WHAMD_PED_XRUN=1): bit-identical, but 8.4 against 6.1 us per launch on a trio at coverage 15 -- the costs of every column formed by
			// every thread are ~100 four-byte loads per thread, where pedslot_run stages the tables once per wave.
			if (run.ncols <= (uint32_t)SLOT_XCOLS && run.n_ends <= (uint32_t)SLOT_XENDS && ex.nf != (uint32_t)PSLOT_FACT4 && debug_env("WHAMD_PED_XRUN") && !debug_env("WHAMD_NO_XRUN")) {
				plan.runs[ri].yflags |= 8u;
				ex.x_off = (uint32_t)(words - (((uint64_t)ex.g_hi << 32) | ex.g_lo));
				words += (pslotx_words(run.ncols, run.threads, run.g) + 3u) & ~(uint64_t)3;
			}
		}
		if (debug_env("WHAMD_PED_XRUN")) words += 65536;   // (an X run requests the cost entries of up to SLOT_XCOLS columns whatever its length: room behind the last run)
		plan.table_words = words;
	}
	// ---- entry / exit layouts.  Exit index of a run in LOGICAL order: bit j = j-th continuing read of its last column.
	plan.f_exit.assign(plan.runs.size(), 0);
	plan.exit_slot.assign(plan.runs.size(), std::vector<uint8_t>());
	parallel_ranges(plan.runs.size(), host_threads(plan.runs.size(), 512), [&](uint64_t r_begin, uint64_t r_end, uint32_t) {
	for (size_t ri = r_begin; ri < r_end; ++ri) {
		const SlotRun& run = plan.runs[ri];
		const uint32_t cl = run.c0 + run.ncols - 1;
		const SlotBtCol& bc = plan.bt_cols[run.row_off + run.ncols - 1];
		for (uint32_t j = 0; j < p.k[cl]; ++j)
			if ((p.fwd_mask[cl] >> j) & 1u) plan.exit_slot[ri].push_back(bc.slot[j]);
		plan.f_exit[ri] = (uint32_t)plan.exit_slot[ri].size();
	}
	});
	// (step si writes the exit side of its own run and the entry side of the next one: steps are independent)
	parallel_ranges(plan.steps.size(), host_threads(plan.steps.size(), 512), [&](uint64_t s_begin, uint64_t s_end, uint32_t) {
	for (size_t si = s_begin; si < s_end; ++si) {
		if (plan.steps[si].kind != 2) continue;
		const uint32_t ri = plan.steps[si].index;
		SlotRun& B = plan.runs[ri];
		const RunDraft& db = drafts[ri];
		// -- entry
		const bool prev_is_run = si > 0 && plan.steps[si - 1].kind == 2 && B.has_prev;
		if (prev_is_run) {
			// filled in when the previous run's exit was linked (below): the exchange layout is chosen per boundary
		} else if (B.has_prev) {
			const ColumnEntry* first = p.col_begin(B.c0);
			std::memset(B.in_pos, 0, sizeof B.in_pos);
			for (uint32_t j = 0; j < p.b[B.c0]; ++j) {
				for (uint32_t s = 0; s < B.L + B.g; ++s)
					if (db.entry_read[s] == (int32_t)first[j].read_id) slot_set_pos(B.in_pos, s, j);
			}
			B.in_fullmask = p.b[B.c0] >= 32 ? 0xFFFFFFFFu : ((1u << p.b[B.c0]) - 1u);
		}
		// -- exit
		const bool next_is_run = si + 1 < plan.steps.size() && plan.steps[si + 1].kind == 2 &&
		                         plan.runs[plan.steps[si + 1].index].has_prev;
		if (next_is_run) {
			// Exchange layout between run B and the next run C: C's grid slots on top (workgroup w of C reads ONE contiguous
			// block), inside the block the reads in the order of the WRITER's physical index -- B's reg slots first, so that
			// the cells of a B thread that differ only in reads local to both runs are adjacent (one 16-byte store instead of
			// four scattered ones), B's lane and wave slots, then B's grid slots with the XCD-selecting bits 0..2 and the
			// halved top slot last (see the grid-slot assignment above).
			SlotRun& C = plan.runs[plan.steps[si + 1].index];
			const RunDraft& dc = drafts[plan.steps[si + 1].index];
			const uint32_t nb = B.L + B.g, nc = C.L + C.g;
			std::vector<std::pair<uint32_t, uint32_t>> local_keys;   // (rank in B, slot in C) of C's local reads
			std::vector<int> slot_in_b(nc, -1);
			for (uint32_t t = 0; t < nc; ++t) {
				if (dc.entry_read[t] < 0) continue;
				for (uint32_t s2 = 0; s2 < nb; ++s2) if (db.exit_read[s2] == dc.entry_read[t]) slot_in_b[t] = (int)s2;
			}
			for (uint32_t t = 0; t < C.L; ++t) {
				if (dc.entry_read[t] < 0) continue;
				const uint32_t s2 = (uint32_t)slot_in_b[t];
				uint32_t key = s2;
				if (s2 >= B.L) {
					const uint32_t gi = s2 - B.L;
					key = (B.g >= 5 && gi + 1 == B.g) ? 3000u : ((B.g >= 5 && gi < 3) ? 2000u + gi : 1000u + gi);
				}
				local_keys.push_back({key, t});
			}
			std::sort(local_keys.begin(), local_keys.end());
			const uint32_t n_loc = (uint32_t)local_keys.size();
			std::memset(C.in_pos, 0, sizeof C.in_pos);
			for (uint32_t i = 0; i < n_loc; ++i) slot_set_pos(C.in_pos, local_keys[i].second, i);
			for (uint32_t i = 0; i < C.g; ++i) slot_set_pos(C.in_pos, C.L + i, n_loc + i);
			C.in_identity = 1;
			for (uint32_t t = 0; t < nc; ++t) if (dc.entry_read[t] >= 0 && slot_pos(C.in_pos, t) != t) C.in_identity = 0;
			C.in_fullmask = (n_loc + C.g) >= 32 ? 0xFFFFFFFFu : ((1u << (n_loc + C.g)) - 1u);
			for (uint32_t t = 0; t < nc; ++t) if (slot_in_b[t] >= 0) slot_set_pos(B.out_pos, (uint32_t)slot_in_b[t], slot_pos(C.in_pos, t));
			B.out_fullmask = C.in_fullmask;
			if (B.half) {
				C.in_half = 1;
				C.in_mirror_pos = slot_pos(B.out_pos, B.L + B.g - 1);   // where this run's top grid read sits in the exchange index
			}
		} else {
			for (uint32_t j = 0; j < plan.f_exit[ri]; ++j) slot_set_pos(B.out_pos, plan.exit_slot[ri][j], j);
			B.out_fullmask = plan.f_exit[ri] >= 32 ? 0xFFFFFFFFu : ((1u << plan.f_exit[ri]) - 1u);
			B.mirror_out = B.half;   // a per-column step reads every entry
		}
	}
	});
	if (getenv("WHAMD_DEBUG_TIMING")) {
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		fprintf(stderr, "[whamd timing] slot plan: setup %.1f ms, column ranges %.1f ms, concatenation %.1f ms, layouts %.1f ms (page faults of this thread: %ld, %ld, %ld, %ld)\n",
		        ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3), ms(tp3, std::chrono::steady_clock::now()), pf1 - pf0, pf2 - pf1, pf3 - pf2, thread_minor_faults() - pf3);
	}
	return true;
}

}  // namespace whamd
