// resident_plan.cpp -- host planner of the forward pass: which columns run as resident segments (resident.h) and
// which go through the per-column kernels; builds every descriptor the resident kernel and the backtrace need.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "resident.h"

namespace whamd {

namespace {

// Fixed-capacity vector on the stack (the planner builds a handful of tiny lists per column: heap allocations were
// half of its time).  Elements beyond the capacity are counted but dropped; every user checks size() against a limit
// far below the capacity.
template <class T, int N>
struct Small {
	T v[N];
	uint32_t n = 0;
	void push_back(const T& x) { if (n < (uint32_t)N) v[n] = x; ++n; }
	size_t size() const { return n; }
	const T* begin() const { return v; }
	const T* end() const { return v + std::min<uint32_t>(n, (uint32_t)N); }
	T& operator[](size_t i) { return v[i]; }
	const T& operator[](size_t i) const { return v[i]; }
};
using Runs = Small<uint32_t, 40>;
using Pairs = Small<std::pair<uint32_t, uint32_t>, 40>;

// Runs of set bits of `mask` as (compact position | mask position << 8 | length << 16); `swap` exchanges the roles
// (deposit: compact -> mask position; extract: mask position -> compact).
uint16_t append_runs(uint32_t mask, bool extract, Runs& out) {
	uint32_t compact = 0;
	uint16_t count = 0;
	for (uint32_t bit = 0; bit < 32;) {
		if (!((mask >> bit) & 1u)) { ++bit; continue; }
		uint32_t len = 0;
		while (bit + len < 32 && ((mask >> (bit + len)) & 1u)) ++len;
		const uint32_t src = extract ? bit : compact, dst = extract ? compact : bit;
		out.push_back(src | (dst << 8) | (len << 16));
		++count;
		compact += len;
		bit += len;
	}
	return count;
}

// Packed runs (source position | destination position << 8 | length << 16) for an arbitrary bit move given as
// (source, destination) pairs sorted by source.
Runs runs_from_pairs(const Pairs& pairs) {
	Runs out;
	for (size_t i = 0; i < pairs.size();) {
		size_t j = i + 1;
		while (j < pairs.size() && pairs[j].first == pairs[j - 1].first + 1 && pairs[j].second == pairs[j - 1].second + 1) ++j;
		out.push_back(pairs[i].first | (pairs[i].second << 8) | ((uint32_t)(j - i) << 16));
		i = j;
	}
	return out;
}

}  // namespace

void plan_forward(const Problem& p, bool resident, int l_pref, bool fold, ResidentPlan& plan, int use_symmetry) {
	const uint32_t n = p.n_cols;
	plan = ResidentPlan();
	plan.col_to_res.assign(n, -1);
	const bool single = p.T == 1 && p.n_ind == 1 && p.value_bound < 1073741824.0;
	const bool ped = p.T == (uint32_t)PED_T && p.n_ind == (uint32_t)PED_NIND && p.n_triples == 1;
	const bool eligible = resident && (single || ped);
	if (ped && l_pref > 7) l_pref = 7;  // trio slices hold T values per entry and every entry is much more work (7 measured best: 8 -> 109 ms, 9 -> 154 ms forward on configs[3])  // trio slices hold T values per entry and every entry is much more work (7 measured best)
	std::vector<uint32_t> last_col;
	if (eligible) {
		last_col.assign(p.n_reads, 0);
		for (uint32_t c = 0; c < n; ++c) {
			const ColumnEntry* col = p.col_begin(c);
			for (uint32_t j = 0; j < p.k[c]; ++j) last_col[col[j].read_id] = c;
		}
	}
	std::vector<uint32_t> ped_abs;  // trio: max over individuals of sum |delta| per column (bound on |L_s|)
	if (eligible && ped) {
		ped_abs.assign(n, 0);
		for (uint32_t cc = 0; cc < n; ++cc) {
			const uint32_t kc = p.k[cc];
			for (uint32_t smp = 0; smp < p.n_ind; ++smp) {
				uint64_t sum = 0;
				for (uint32_t j = 0; j < kc; ++j) sum += (uint64_t)std::llabs((long long)p.delta[(size_t)p.col_ptr[cc] * p.n_ind + (size_t)smp * kc + j]);
				ped_abs[cc] = (uint32_t)std::max<uint64_t>(ped_abs[cc], std::min<uint64_t>(sum, 0xFFFFFFFFull));
			}
		}
	}
	// per run: which bits of the entering / exit logical index are grid reads (for the exchange layouts below)
	std::vector<std::vector<uint8_t>> entry_grid, exit_grid;
	uint32_t c = 0;
	while (c < n) {
		if (!eligible) {
			plan.steps.push_back(Step{0, c});
			++c;
			continue;
		}
		const uint32_t b0 = p.b[c];
		const ColumnEntry* first = p.col_begin(c);
		uint32_t g = 0;
		// one workgroup per CU (256 = 2^8) as long as the slice stays <= 2^12 entries; more workgroups only for the
		// highest coverages (22, 23), where the slice would not fit otherwise
		// with the complement symmetry only half of the workgroups are launched: one more grid read keeps 2^8 of them busy
		// (only where the chip would otherwise be full: below that, a run is bound by latency and by its fixed cost, and
		// one more grid read only shortens it -- measured at coverage 12..18)
		const bool try_half = single && use_symmetry > 0;
		const bool bump = try_half && (int)b0 - l_pref >= 8;
		const int lp = bump ? l_pref - 1 : l_pref;
		if ((int)b0 > lp) g = std::min<uint32_t>(b0 - (uint32_t)lp, bump ? 9u : 8u);
		if (b0 - g > 12) g = std::min<uint32_t>(b0 - 12, RES_GMAX);
		// grid reads: the g entering reads that end last (ties: the younger read)
		std::vector<uint32_t> order(b0);
		for (uint32_t j = 0; j < b0; ++j) order[j] = j;
		std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t bb) {
			const uint32_t ea = last_col[first[a].read_id], eb = last_col[first[bb].read_id];
			if (ea != eb) return ea > eb;
			return a > bb;
		});
		std::vector<uint32_t> grid_reads;
		uint32_t grid_end = 0xFFFFFFFFu;
		for (uint32_t i = 0; i < g; ++i) {
			grid_reads.push_back(first[order[i]].read_id);
			grid_end = std::min(grid_end, last_col[first[order[i]].read_id]);
		}
		std::sort(grid_reads.begin(), grid_reads.end());
		uint32_t c1 = c, run_max_l = 0;
		uint64_t run_stage = 0, run_terms = 0;
		while (c1 < n && c1 - c < (uint32_t)RES_MAXCOLS) {
			if (c1 + 1 == n) break;      // the last column needs the global optimum
			if (c1 >= grid_end) break;   // a grid read is minimised out at this column
			if (single && c1 > c && p.b[c1] == 0) break;  // next connected component: its own run (and maybe its own stream)
			const uint32_t kc = p.k[c1];
			if (kc < g) break;
			const uint32_t Lb = p.b[c1] - g, Lf = p.f[c1] - g, Lk = kc - g;
			if (kc - p.f[c1] > (uint32_t)RES_EMAX) break;
			if (ped) {
				if (Lb > (uint32_t)PED_LMAX || Lf > (uint32_t)PED_LMAX || Lk > (uint32_t)PED_LKMAX) break;
				run_max_l = std::max(run_max_l, std::max(Lb, Lf));
				run_stage += ((1ull << Lf) + 1) / 2;  // one u32 per projection entry
				run_terms += p.term_end(c1, p.T - 1) - p.term_begin(c1, 0);
				const uint64_t lds_bytes = (uint64_t)(c1 - c + 1) * (PED_LDSWORDS + PED_TABLE) * 4 + run_terms * 8 + 2 * (16ull << run_max_l) + run_stage * 8 + 16;
				if (lds_bytes > 150 * 1024) break;
				if (ped_abs[c1] >= (1u << 23)) break;  // L_s must fit the 24-bit multiply-add of the term evaluation
			} else {
				if (Lb > (uint32_t)RES_LMAX || Lf > (uint32_t)RES_LMAX || Lk > 14) break;
				run_max_l = std::max(run_max_l, std::max(Lb, Lf));
				run_stage += (uint64_t)(kc - p.f[c1]) * std::max<uint32_t>(4, (1u << Lf) / 32);
				const uint64_t lds_bytes = (uint64_t)(c1 - c + 1) * (64 + RES_TABLE) * 4 + 2 * (4ull << run_max_l) + run_stage * 8;
				if (lds_bytes > 150 * 1024) break;
			}
			++c1;
		}
		if (c1 - c < 2) {
			plan.steps.push_back(Step{0, c});
			++c;
			continue;
		}
		// ---- emit the segment [c, c1)
		ResSegment seg{};
		seg.c0 = c;
		seg.ncols = c1 - c;
		seg.g = g;
		seg.col_off = (uint32_t)plan.columns.size();
		seg.has_prev = c > 0;
		seg.kind = ped ? 1u : 0u;
		seg.term_off = (uint32_t)plan.ped_terms.size();
		uint32_t max_l = 0, stage_words = 0;
		std::vector<uint8_t> run_entry_grid, run_exit_grid;
		bool out_ok = true, bt_ok = true;
		const size_t columns_mark = plan.columns.size();
		auto is_grid = [&](uint32_t read) { return std::binary_search(grid_reads.begin(), grid_reads.end(), read); };
		{   // load layout: positions in the entering index == positions in column c (shared reads are its low bits)
			uint32_t gm = 0;
			for (uint32_t j = 0; j < b0; ++j) if (is_grid(first[j].read_id)) gm |= 1u << j;
			const uint32_t lm = (b0 >= 32 ? 0xFFFFFFFFu : ((1u << b0) - 1u)) & ~gm;
			run_entry_grid.assign(b0, 0);
			for (uint32_t j = 0; j < b0; ++j) run_entry_grid[j] = (gm >> j) & 1u;
			Runs rg, rl;
			seg.n_in_grid = append_runs(gm, false, rg);
			seg.n_in_local = append_runs(lm, false, rl);
			if (rg.size() > (size_t)RES_IOSEG || rl.size() > (size_t)RES_IOSEG) {  // exotic layout: leave this column to the column kernels
				plan.steps.push_back(Step{0, c});
				++c;
				continue;
			}
			std::copy(rg.begin(), rg.end(), seg.in_grid);
			std::copy(rl.begin(), rl.end(), seg.in_local);
			seg.Lb0 = b0 - g;
		}
		for (uint32_t cc = c; cc < c1; ++cc) {
			const ColumnEntry* col = p.col_begin(cc);
			const uint32_t kc = p.k[cc];
			const int32_t* dl = p.delta.data() + (size_t)p.col_ptr[cc];  // n_ind == 1
			ResColumn rc{};
			rc.Lb = p.b[cc] - g;
			rc.Lf = p.f[cc] - g;
			rc.ebits = kc - p.f[cc];
			max_l = std::max(max_l, std::max(rc.Lb, rc.Lf));
			rc.Cp = rc.Cm = RES_ABSENT;
			rc.Cc = INF;
			for (uint64_t q = p.term_begin(cc, 0); q < p.term_end(cc, 0); ++q) {
				const CostTerm& t = p.terms[q];
				if (t.plus) rc.Cp = t.c;
				else if (t.minus) rc.Cm = t.c;
				else rc.Cc = std::min(rc.Cc, t.c);
			}
			// logical -> (grid slot | local bit)
			uint32_t li = 0, gi = 0;
			int local_of[40], grid_of[40];
			for (uint32_t j = 0; j < kc && j < 40; ++j) local_of[j] = grid_of[j] = -1;
			for (uint32_t j = 0; j < kc; ++j) {
				if (is_grid(col[j].read_id)) {
					grid_of[j] = (int)gi;
					rc.dgrid[gi++] = dl[j];
				} else {
					local_of[j] = (int)li;
					rc.dloc[li] = dl[j];
					++li;
				}
			}
			uint32_t en = 0;
			for (uint32_t j = 0; j < kc; ++j) {  // ending reads in ascending logical position
				if (local_of[j] < 0 || ((p.fwd_mask[cc] >> j) & 1u)) continue;
				uint32_t mg = 0, ml = 0;
				for (uint32_t q = j + 1; q < kc; ++q) {
					if (grid_of[q] >= 0) mg |= 1u << grid_of[q]; else ml |= 1u << local_of[q];
				}
				rc.epos[en] = (uint32_t)local_of[j];
				rc.mG[en] = mg;
				rc.mL[en] = ml;
				++en;
			}
			rc.d0 = rc.dloc[0];
			rc.d1 = rc.dloc[1];
			rc.d2 = rc.dloc[2];
			rc.dE = rc.ebits ? rc.dloc[rc.epos[0]] : 0;
			rc.lowmask = (1u << rc.Lb) - 1u;
			rc.nthr = (1u << rc.Lf) >> 2;
			{   // packed 16-bit evaluation words
				const uint32_t s01 = (uint32_t)rc.d0, s2 = (uint32_t)rc.d1, s4 = (uint32_t)rc.d2;
				auto pack = [](uint32_t lo, uint32_t hi) { return (lo & 0xFFFFu) | (hi << 16); };
				rc.pk[0] = pack(0, s01);
				rc.pk[1] = pack(s2, s2 + s01);
				rc.pk[2] = pack(s4, s4 + s01);
				rc.pk[3] = pack(s4 + s2, s4 + s2 + s01);
				const uint32_t K = rc.Cp + rc.Cm, cc = std::min<uint32_t>(rc.Cc, 0xFFFFu);
				rc.A = rc.Cp;
				rc.Kpk = pack(K, K);
				rc.Ccpk = pack(cc, cc);
				rc.pk_ok = (rc.Cp != RES_ABSENT && rc.Cm != RES_ABSENT && rc.Cp < (1u << 14) && rc.Cm < (1u << 14) && K < (1u << 14)) ? 1u : 0u;
				rc.ep0 = rc.epos[0];
				rc.mL0 = rc.mL[0];
			}
			// vectorised path: a thread owns 4 consecutive projection entries (8 cells when a read ends) and moves them
			// with 16-byte LDS accesses; needs aligned groups in the previous slice
			const bool fast = rc.ebits <= 1 && rc.Lf >= 2 && rc.Lb >= 3;
			if (!fast) rc.mode = RES_MODE_GENERIC;
			else if (rc.ebits == 0) rc.mode = RES_MODE_E0;
			else rc.mode = rc.epos[0] >= 2 ? RES_MODE_E1_HIGH : (rc.epos[0] == 0 ? RES_MODE_E1_BIT0 : RES_MODE_E1_BIT1);
			rc.pbits = 0;
			for (uint32_t u = 0; u < 4; ++u) {  // low bits of the side-0 cell of entry 4t+u (dp_device.hip res_finish_entries)
				const uint32_t low = rc.mode == RES_MODE_E1_HIGH ? u : (rc.mode == RES_MODE_E1_BIT0 ? 2 * u : (((u >> 1) << 2) | (u & 1)));
				rc.pbits |= ((uint32_t)__builtin_popcount(low & rc.mL[0]) & 1u) << u;
			}
			rc.pbits |= (((uint32_t)__builtin_popcount(rc.mG[0]) + (uint32_t)__builtin_popcount(rc.mL[0])) & 1u) << 8;
			// record of a vectorised column: one byte per thread (nthr bytes); otherwise ballot words per plane
			rc.nwords = fast ? std::max<uint32_t>(1, rc.nthr / 8) : std::max<uint32_t>(1, (1u << rc.Lf) / 64);
			if (ped) {
				// one u32 per projection entry on EVERY column (the transmission argmin is needed everywhere)
				rc.mode = RES_MODE_GENERIC;
				rc.nwords = ((1u << rc.Lf) + 1) / 2;
				rc.stage_off = stage_words;
				stage_words += rc.nwords;
				PedColumn pc{};
				pc.Lb = rc.Lb; pc.Lf = rc.Lf; pc.ebits = rc.ebits; pc.stage_off = rc.stage_off * 2;
				pc.lowmask = rc.lowmask; pc.recomb = p.recomb[cc];
				for (uint32_t q = 0; q < 4; ++q) { pc.epos[q] = rc.epos[q]; pc.mL[q] = rc.mL[q]; pc.mG[q] = rc.mG[q]; }
				for (uint32_t j = 0; j < kc; ++j) {
					const uint32_t smp = col[j].sample;
					const int32_t dj = p.delta[(size_t)p.col_ptr[cc] * p.n_ind + (size_t)smp * kc + j];
					if (grid_of[j] >= 0) pc.dgrid[smp][grid_of[j]] = dj; else pc.dloc[smp][local_of[j]] = dj;
				}
				pc.term_off = (uint32_t)plan.ped_terms.size() - seg.term_off;
				const uint64_t tb = p.term_begin(cc, 0);
				for (uint32_t t = 0; t <= p.T; ++t) pc.tptr[t] = (uint32_t)((t < p.T ? p.term_begin(cc, t) : p.term_end(cc, p.T - 1)) - tb);
				pc.n_terms = pc.tptr[p.T];
				for (uint32_t t = 0; t < p.T; ++t) {
					pc.maxcnt = std::max(pc.maxcnt, pc.tptr[t + 1] - pc.tptr[t]);
					for (uint32_t k = 0; k < (uint32_t)PED_REGTERMS; ++k) pc.rterms[t][k] = PedTerm{0xFFFFFFFFu, 0u};
				}
				for (uint32_t q = 0; q < rc.ebits; ++q)
					for (uint32_t smp = 0; smp < p.n_ind; ++smp) pc.dE[q][smp] = pc.dloc[smp][rc.epos[q]];
				for (uint64_t q = tb; q < tb + pc.n_terms; ++q) {
					uint32_t sig = 0;
					for (uint32_t smp = 0; smp < p.n_ind; ++smp) {
						const int sg1 = (int)((p.terms[q].plus >> smp) & 1u) - (int)((p.terms[q].minus >> smp) & 1u);
						sig |= ((uint32_t)sg1 & 0xFFu) << (8 * smp);
					}
					plan.ped_terms.push_back(PedTerm{p.terms[q].c, sig});
					for (uint32_t t = 0; t < p.T; ++t) {
						const uint64_t rel = q - tb;
						if (rel >= pc.tptr[t] && rel < pc.tptr[t + 1] && rel - pc.tptr[t] < (uint64_t)PED_REGTERMS)
							pc.rterms[t][rel - pc.tptr[t]] = PedTerm{p.terms[q].c, sig};
					}
				}
				plan.ped_columns.resize(plan.columns.size() + 1);
				plan.ped_columns.back() = pc;
			} else {
				rc.stage_off = stage_words;
				stage_words += rc.ebits * rc.nwords;
			}
			// backtrace record (resident.h): local-space walk + logical index from (workgroup index, local cell index)
			ResBacktrace rb{};
			uint32_t gmf = 0, fi = 0;
			for (uint32_t j = 0; j < kc; ++j) {
				if (!((p.fwd_mask[cc] >> j) & 1u)) continue;
				if (grid_of[j] >= 0) gmf |= 1u << fi;
				++fi;
			}
			const uint32_t lmf = (fi >= 32 ? 0xFFFFFFFFu : ((1u << fi) - 1u)) & ~gmf;
			{
				Pairs gp, lp;
				for (uint32_t j = 0; j < kc; ++j) {
					if (grid_of[j] >= 0) gp.push_back({(uint32_t)grid_of[j], j}); else lp.push_back({(uint32_t)local_of[j], j});
				}
				const Runs gr = runs_from_pairs(gp), lr = runs_from_pairs(lp);
				if (gr.size() > (size_t)RES_BT_GRUNS || lr.size() > (size_t)RES_BT_LRUNS) bt_ok = false;
				else {
					rb.n_g = (uint32_t)gr.size();
					rb.n_l = (uint32_t)lr.size();
					std::copy(gr.begin(), gr.end(), rb.gruns);
					std::copy(lr.begin(), lr.end(), rb.lruns);
				}
			}
			rb.Lf = rc.Lf;
			rb.ebits = rc.ebits;
			rb.nwords = rc.nwords;
			rb.layout = ped ? 2u : (fast ? 1u : 0u);
			rb.stage_off = rc.stage_off;
			for (uint32_t q = 0; q < 3; ++q) rb.epos[q] = rc.epos[q];
			if (cc + 1 == c1) {  // store layout of the exit state
				run_exit_grid.assign(fi, 0);
				for (uint32_t j = 0; j < fi; ++j) run_exit_grid[j] = (gmf >> j) & 1u;
				Runs rg, rl;
				seg.n_out_grid = append_runs(gmf, false, rg);
				seg.n_out_local = append_runs(lmf, false, rl);
				out_ok = rg.size() <= (size_t)RES_IOSEG && rl.size() <= (size_t)RES_IOSEG;
				if (out_ok) {
					std::copy(rg.begin(), rg.end(), seg.out_grid);
					std::copy(rl.begin(), rl.end(), seg.out_local);
					Runs we, le;
					seg.n_wext = append_runs(gmf, true, we);
					std::copy(we.begin(), we.end(), seg.wext);
					seg.n_lext = append_runs(lmf, true, le);
					if (le.size() > 10) out_ok = false; else std::copy(le.begin(), le.end(), seg.lext);
				}
				seg.Lf_last = rc.Lf;
			}
			plan.col_to_res[cc] = (int32_t)plan.columns.size();
			plan.columns.push_back(rc);
			plan.backtrace.push_back(rb);
		}
		// fold columns in which no read ends into the next vectorised column (resident.h RES_MODE_FOLDED)
		if (fold && !ped) {
			uint32_t run = 0;
			for (size_t i = columns_mark; i < plan.columns.size(); ++i) {
				ResColumn& rc = plan.columns[i];
				const bool has_next = i + 1 < plan.columns.size();
				if (rc.mode == RES_MODE_E0 && has_next && plan.columns[i + 1].mode != RES_MODE_GENERIC && run < RES_MAXFOLD) {
					rc.mode = RES_MODE_FOLDED;
					++run;
				} else {
					if (run) {
						rc.nfold = run;
						rc.lowmask = plan.columns[i - run].lowmask;  // the slice read is the one the first folded column would have read
						for (uint32_t f = 1; f <= run; ++f) rc.pk_ok &= plan.columns[i - f].pk_ok;
					}
					run = 0;
				}
			}
		}
		{   // complement symmetry: every column vectorised and cost(~x) == cost(x), i.e. Cp + (sum of all deltas) == Cm
			bool half = try_half && g >= (use_symmetry > 1 ? 1u : 8u);  // below 2^8 a run is latency-bound: nothing to gain
			for (size_t i = columns_mark; half && i < plan.columns.size(); ++i) {
				const ResColumn& rc = plan.columns[i];
				if (rc.mode == RES_MODE_GENERIC) half = false;
				uint32_t sum = 0;
				for (uint32_t q = 0; q < g; ++q) sum += (uint32_t)rc.dgrid[q];
				for (uint32_t q = 0; q < 14; ++q) sum += (uint32_t)rc.dloc[q];
				const bool both_absent = rc.Cp == RES_ABSENT && rc.Cm == RES_ABSENT;
				if (!both_absent && (rc.Cp == RES_ABSENT || rc.Cm == RES_ABSENT || rc.Cp + sum != rc.Cm)) half = false;
			}
			seg.half = half ? 1u : 0u;
		}
		for (size_t i = columns_mark; i < plan.columns.size(); ++i) {
			ResColumn& rc = plan.columns[i];
			rc.flags = rc.mode | (rc.nfold << 8) | (rc.pk_ok << 12);
		}
		if (!out_ok || !bt_ok) {  // exotic layout: undo and leave the first column to the column kernels
			for (uint32_t cc = c; cc < c1; ++cc) plan.col_to_res[cc] = -1;
			plan.columns.resize(columns_mark);
			plan.backtrace.resize(columns_mark);
			if (plan.ped_columns.size() > columns_mark) plan.ped_columns.resize(columns_mark);
			plan.ped_terms.resize(seg.term_off);
			plan.steps.push_back(Step{0, c});
			++c;
			continue;
		}
		// 512 threads per workgroup: a thread of the vectorised path then owns two groups of 4 entries (same speed as 1024
		// threads for one table at 2^12-entry slices: the step is VALU-bound either way; smaller slices keep one group per thread), but two workgroups of DIFFERENT tables fit on a CU,
		// which 2 x 16 waves never did: independent blocks in flight overlap (2 blocks: 1.07 -> 1.46 M columns/s)
		seg.threads = ped ? std::min<uint32_t>(512, std::max<uint32_t>(64, 4u << max_l))
		                  : std::min<uint32_t>(512, std::max<uint32_t>(64, (1u << max_l) / 4));
		seg.n_terms = (uint32_t)plan.ped_terms.size() - seg.term_off;
		seg.max_l = max_l;
		seg.stage_words = stage_words;
		{   // backtrace chain of the run (resident.h ResBacktrace)
			ResBacktrace* rb = plan.backtrace.data() + columns_mark;
			const uint32_t nc = seg.ncols;
			uint32_t k = 0, src = RES_BT_NONE, cm = 0xFFFFFFFFu;
			bool simple = !ped;  // trio records carry the transmission argmin on every column: general walk
			for (uint32_t ci = nc; ci-- > 0;) {
				cm &= (1u << rb[ci].Lf) - 1u;
				rb[ci].src = src;
				rb[ci].cmask = cm;
				rb[ci].kpos = RES_BT_NONE;
				if (rb[ci].ebits) {
					rb[ci].kpos = k;
					rb[k].kcol = ci;
					++k;
					src = ci;
					cm = 0xFFFFFFFFu;
					simple = simple && rb[ci].layout == 1u && rb[ci].ebits == 1u;
				}
			}
			seg.bt_active = (uint16_t)k;
			bool ped_simple = ped;  // trio: every column has a one-byte-per-(entry, value) record and at most one ending read
			for (uint32_t ci = 0; ci < nc; ++ci) ped_simple = ped_simple && rb[ci].layout == 2u && rb[ci].ebits <= 1u;
			seg.bt_simple = simple ? 1 : (ped_simple ? 2 : 0);
		}

		plan.steps.push_back(Step{1, (uint32_t)plan.segments.size()});
		plan.segments.push_back(seg);
		entry_grid.push_back(run_entry_grid);
		exit_grid.push_back(run_exit_grid);
		plan.n_resident_columns += seg.ncols;
		c = c1;
	}
	if (single) {
		for (size_t si = 0; si < plan.steps.size(); ++si) {
			const uint32_t c0 = plan.steps[si].kind == 1 ? plan.segments[plan.steps[si].index].c0 : plan.steps[si].index;
			if (si == 0 || p.b[c0] == 0) plan.component_first_step.push_back((uint32_t)si);
		}
	}
	// ---- exchange layouts between consecutive runs.  A re-layout is an all-to-all between workgroups; in logical
	// order the writer of run A scatters 4..16-byte pieces (its grid reads sit in the middle of the index), which costs
	// ~9 us per boundary on MI355X.  Instead the slice is stored as [grid reads of B | grid reads of A | bits local in
	// both], so A writes 2^(shared bits) contiguous entries per destination block and B reads one contiguous block.
	for (size_t si = 0; si + 1 < plan.steps.size(); ++si) {
		if (plan.steps[si].kind != 1 || plan.steps[si + 1].kind != 1) continue;
		ResSegment& A = plan.segments[plan.steps[si].index];
		ResSegment& B = plan.segments[plan.steps[si + 1].index];
		const std::vector<uint8_t>& ga = exit_grid[plan.steps[si].index];
		const std::vector<uint8_t>& gb = entry_grid[plan.steps[si + 1].index];
		if (ga.size() != gb.size()) continue;
		const uint32_t f = (uint32_t)ga.size();
		uint32_t pi[40] = {0};
		uint32_t next = 0;
		for (uint32_t j = 0; j < f; ++j) if (!ga[j] && !gb[j]) pi[j] = next++;
		for (uint32_t j = 0; j < f; ++j) if (ga[j] && !gb[j]) pi[j] = next++;
		for (uint32_t j = 0; j < f; ++j) if (gb[j]) pi[j] = next++;
		Pairs aw, al, bw, bl;
		uint32_t sa = 0, la = 0, sb = 0, lb = 0;
		for (uint32_t j = 0; j < f; ++j) {
			if (ga[j]) aw.push_back({sa++, pi[j]}); else al.push_back({la++, pi[j]});
			if (gb[j]) bw.push_back({sb++, pi[j]}); else bl.push_back({lb++, pi[j]});
		}
		const Runs raw = runs_from_pairs(aw), ral = runs_from_pairs(al), rbw = runs_from_pairs(bw), rbl = runs_from_pairs(bl);
		if (raw.size() > (size_t)RES_IOSEG || ral.size() > (size_t)RES_IOSEG || rbw.size() > (size_t)RES_IOSEG || rbl.size() > (size_t)RES_IOSEG) continue;
		A.n_out_grid = (uint16_t)raw.size();
		A.n_out_local = (uint16_t)ral.size();
		std::copy(raw.begin(), raw.end(), A.out_grid);
		std::copy(ral.begin(), ral.end(), A.out_local);
		B.n_in_grid = (uint16_t)rbw.size();
		B.n_in_local = (uint16_t)rbl.size();
		std::copy(rbw.begin(), rbw.end(), B.in_grid);
		std::copy(rbl.begin(), rbl.end(), B.in_local);
	}
	// ---- complement symmetry across step boundaries: who reads what a halved run wrote
	for (size_t si = 0; si < plan.steps.size(); ++si) {
		if (plan.steps[si].kind != 1) continue;
		ResSegment& A = plan.segments[plan.steps[si].index];
		const uint32_t f_exit = A.Lf_last + A.g;
		A.out_fullmask = f_exit >= 32 ? 0xFFFFFFFFu : ((1u << f_exit) - 1u);
		if (!A.half) continue;
		const bool next_is_run = si + 1 < plan.steps.size() && plan.steps[si + 1].kind == 1;
		if (!next_is_run) { A.mirror_out = 1; continue; }
		// position of A's top grid-read bit in the index A stores with (wout | deposit(l))
		uint32_t bit = 0xFFFFFFFFu;
		for (uint32_t r = 0; r < A.n_out_grid; ++r) {
			const uint32_t run = A.out_grid[r], compact = run & 255u, pos = (run >> 8) & 255u, len = run >> 16;
			if (A.g - 1 >= compact && A.g - 1 < compact + len) bit = pos + (A.g - 1 - compact);
		}
		if (bit == 0xFFFFFFFFu) { A.mirror_out = 1; continue; }
		ResSegment& B = plan.segments[plan.steps[si + 1].index];
		B.in_half = 1;
		B.in_mirror_bit = bit;
		B.in_fullmask = A.out_fullmask;
	}
}

}  // namespace whamd
