// pedslot_emulate.cpp -- HOST-ONLY DIAGNOSTIC of the pedigree slot runs (whamd_debug_emulate_slot_plan on a table with
// T > 1): executes a pedigree SlotPlan the way kernels_pedslots.h / kernels_backtrace.h do -- the same tables (G, W, S of
// slots.h), one (cell, transmission value) per lane, the butterfly min-plus step with its tie rule, the pair decisions of
// the ending reads, one record byte per lane and column, the column-by-column walk -- so that the CPU test-suite can check
// the PLAN and the record semantics against the oracle without a GPU.  Exponential-size bookkeeping for small inputs, not a
// solver: no product path calls it.
//
// Reference semantics: src/pedigreedptable.cpp:240-327 (cost, min over the previous transmission value with the lowest j
// on ties, strict-'<' projection in Gray-code order), :137-173 (backtrace).
#include <algorithm>
#include <cstring>

#include "slots.h"

namespace whamd {

namespace {

inline uint32_t gray_rank_host(uint32_t x) {
	uint32_t r = x;
	for (uint32_t s = 1; s < 32; s <<= 1) r ^= r >> s;
	return r;
}
inline uint32_t bit(uint32_t v, uint32_t s) { return (v >> s) & 1u; }
inline uint32_t sat_add_host(uint32_t a, uint32_t b) { const uint64_t s = (uint64_t)a + b; return s > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)s; }

// cost of cell x of column c for transmission value t from the host term list
uint32_t cell_cost_ped(const Problem& p, uint32_t c, uint32_t t, uint32_t x) {
	const uint32_t k = p.k[c];
	const int32_t* dl = p.delta.data() + (size_t)p.col_ptr[c] * p.n_ind;
	uint32_t best = INF;
	for (uint64_t q = p.term_begin(c, t); q < p.term_end(c, t); ++q) {
		const CostTerm& tm = p.terms[q];
		uint32_t v = tm.c;
		for (uint32_t s = 0; s < p.n_ind; ++s) {
			if (!(bit(tm.plus, s) | bit(tm.minus, s))) continue;
			uint32_t Ls = 0;
			for (uint32_t j = 0; j < k; ++j) if (bit(x, j)) Ls += (uint32_t)dl[(size_t)s * k + j];
			if (bit(tm.plus, s)) v += Ls; else v -= Ls;
		}
		best = std::min(best, v);
	}
	return best;
}

}  // namespace

uint32_t pedslot_table_entry(const Problem& p, const SlotPlan& plan, uint32_t run_index, int kind, uint32_t unit, uint32_t c, uint32_t t, uint32_t f) {
	const SlotRun& run = plan.runs[run_index];
	const PedSlotExtra& ex = plan.pextra[run_index];
	const PedSlotRow& row = plan.prows[run.row_off + c];
	const uint32_t col = run.c0 + c;
	const bool fact4 = ex.nf == (uint32_t)PSLOT_FACT4;   // a quartet's line: per column, the same for every transmission value
	const bool fact = ex.nf == (uint32_t)PSLOT_FACT || fact4;   // entry f of the factorised line (Problem::fterms): always present
	const uint64_t q = p.term_begin(col, t) + f;
	if (!fact && q >= p.term_end(col, t)) return kind == 0 ? INF : 0u;   // absent form: INF + 0 + 0
	const CostTerm& tm = fact4 ? p.fterms[(size_t)col * PSLOT_FSTRIDE4 + f] : (fact ? p.fterms[((size_t)col * p.T + t) * 16 + f] : p.terms[q]);
	const uint32_t nls = 6u - ex.tb;
	uint32_t acc = kind == 0 ? tm.c : 0u;
	uint32_t s0, s1, bits;
	if (kind == 0) { s0 = run.L; s1 = run.L + run.g; bits = unit; }
	else if (kind == 1) { s0 = nls; s1 = run.L; bits = unit; }
	else { s0 = 0; s1 = nls; bits = unit >> ex.tb; }
	for (uint32_t s = s0; s < s1; ++s) {
		if (!bit(bits, s - s0)) continue;
		const uint32_t ind = row.ind[s];
		if (bit(tm.plus, ind)) acc += (uint32_t)row.dslot[s];
		else if (bit(tm.minus, ind)) acc -= (uint32_t)row.dslot[s];
	}
	return acc;
}

bool emulate_pedslot_plan(const Problem& p, const SlotPlan& plan, std::vector<uint32_t>& path_index, std::vector<uint32_t>& path_trans,
                          uint32_t& score, std::string& msg) {
	const uint32_t n = p.n_cols, T = p.T;
	path_index.assign(n, 0);
	path_trans.assign(n, 0);
	score = 0;
	if (n == 0) return true;
	if (!plan.ped) { msg = "not a pedigree plan"; return false; }
	std::vector<uint32_t> pr, nx;                          // exchange buffers: [index][T]
	std::vector<std::vector<uint32_t>> col_arg(n);          // per-column steps: [entry * T + t] argmin cell | argj << 28
	std::vector<std::vector<uint8_t>> records(plan.runs.size());
	uint32_t last_x = 0, last_t = 0, last_j = 0, total = INF;
	for (size_t si = 0; si < plan.steps.size(); ++si) {
		const Step& st = plan.steps[si];
		if (st.kind == 0) {
			const uint32_t c = st.index, k = p.k[c], b = p.b[c], f = p.f[c];
			const bool is_last = c + 1 == n;
			nx.assign(((size_t)1 << f) * T, INF);
			col_arg[c].assign(((size_t)1 << f) * T, 0);
			std::vector<uint32_t> best_rank(((size_t)1 << f) * T, 0xFFFFFFFFu);
			uint32_t opt = INF, opt_rank = 0xFFFFFFFFu;
			for (uint32_t x = 0; x < (1u << k); ++x) {
				const uint32_t rank = gray_rank_host(x);
				uint32_t y = 0, fi = 0;
				for (uint32_t j = 0; j < k; ++j) if (bit(p.fwd_mask[c], j)) y |= bit(x, j) << fi++;
				for (uint32_t i = 0; i < T; ++i) {
					const uint32_t cost = cell_cost_ped(p, c, i, x);
					uint32_t mn = INF, mj = 0;
					for (uint32_t j = 0; j < T; ++j) {
						const uint32_t prev = c == 0 ? 0u : pr[(size_t)(x & ((1u << b) - 1u)) * T + j];
						uint32_t val = (cost != INF && prev != INF) ? cost + prev : INF;
						if (val != INF) val += (uint32_t)__builtin_popcount(i ^ j) * p.recomb[c];
						if (val < mn) { mn = val; mj = j; }
					}
					if (is_last) {
						if (mn < opt || (mn == opt && mn != INF && rank < opt_rank)) { opt = mn; opt_rank = rank; last_x = x; last_t = i; last_j = mj; }
						continue;
					}
					const size_t e = (size_t)y * T + i;
					if (mn < nx[e] || (mn == nx[e] && mn != INF && rank < best_rank[e])) { nx[e] = mn; best_rank[e] = rank; col_arg[c][e] = x | (mj << 28); }
				}
			}
			if (is_last) total = opt;
			pr.swap(nx);
			continue;
		}
		const SlotRun& run = plan.runs[st.index];
		const PedSlotExtra& ex = plan.pextra[st.index];
		if (run.lr != 0 || run.half || (1u << ex.tb) != T) { msg = "pedigree run with reg slots / halved / wrong T"; return false; }
		const uint32_t TB = ex.tb, NLS = 6u - TB, L = run.L, nslots = L + run.g, nwg = 1u << run.g, threads = run.threads;
		if (L != NLS + run.lw || threads != (64u << run.lw)) { msg = "pedigree run: L / threads inconsistent"; return false; }
		const uint32_t ncell = nwg << L;
		std::vector<uint32_t> D((size_t)ncell * T, 0), V((size_t)ncell * T), J((size_t)ncell * T), V2, J2;
		records[st.index].assign((size_t)nwg * ex.rec_words * 4u, 0);
		if (run.has_prev) {
			for (uint32_t P = 0; P < ncell; ++P) {
				uint32_t idx = 0;
				if (run.in_identity) idx = P & run.in_occ;
				else for (uint32_t s = 0; s < nslots; ++s) if (bit(run.in_occ, s)) idx |= bit(P, s) << slot_pos(run.in_pos, s);
				if ((size_t)idx * T + T > pr.size()) { msg = "pedigree run reads beyond the exchange buffer"; return false; }
				for (uint32_t t = 0; t < T; ++t) D[(size_t)P * T + t] = pr[(size_t)idx * T + t];
			}
		}
		for (uint32_t ci = 0; ci < run.ncols; ++ci) {
			const PedSlotRow& row = plan.prows[run.row_off + ci];
			// cost + min-plus butterfly, lane by lane
			for (uint32_t P = 0; P < ncell; ++P) {
				for (uint32_t t = 0; t < T; ++t) { V[(size_t)P * T + t] = D[(size_t)P * T + t]; J[(size_t)P * T + t] = t; }
			}
			for (uint32_t s = 0; s < TB; ++s) {
				V2 = V; J2 = J;
				for (uint32_t P = 0; P < ncell; ++P)
					for (uint32_t t = 0; t < T; ++t) {
						const size_t me = (size_t)P * T + t, other = (size_t)P * T + (t ^ (1u << s));
						const uint32_t cand = sat_add_host(V[other], row.recomb);
						if (cand < sat_add_host(V[me], bit(t, s))) { V2[me] = cand; J2[me] = J[other]; }
					}
				V.swap(V2); J.swap(J2);
			}
			for (uint32_t P = 0; P < ncell; ++P) {
				const uint32_t w = P >> L, l = P & ((1u << L) - 1u), wave = l >> NLS;
				for (uint32_t t = 0; t < T; ++t) {
					const uint32_t tid = (l << TB) | t, lane = tid & 63u;
					uint32_t cost = INF;
					if (ex.nf == (uint32_t)PSLOT_FACT) {   // the factorised line, operation by operation as pedslot_run_body
						uint32_t a[16], sl[4];
						for (uint32_t f = 0; f < 4; ++f)
							a[f] = pedslot_table_entry(p, plan, st.index, 0, w, ci, t, f) + pedslot_table_entry(p, plan, st.index, 1, wave, ci, t, f);
						for (uint32_t f = 4; f < 16; ++f) a[f] = p.fterms[((size_t)(run.c0 + ci) * T + t) * 16 + f].c;   // K: the same for every cell
						for (uint32_t f = 0; f < 4; ++f) sl[f] = pedslot_table_entry(p, plan, st.index, 2, lane, ci, t, f);
						const uint32_t X = a[0] + sl[0], Y = a[1] + sl[1], C = a[2] + sl[2];
						const uint32_t M0 = std::min(a[4], a[5] + X), M1 = std::min(a[6], a[7] - X);
						const uint32_t F0 = std::min(a[8], a[9] + Y), F1 = std::min(a[10], a[11] - Y);
						cost = std::min(std::min(a[12] + M0 + F0, a[13] + C + M0 + F1), std::min(a[14] - C + M1 + F0, a[15] + M1 + F1));
					} else if (ex.nf == (uint32_t)PSLOT_FACT4) {   // a quartet's factorised line: the tables per column, the wiring from the lane's transmission value
						uint32_t L4[4], k[16];
						for (uint32_t f = 0; f < 4; ++f)
							L4[f] = pedslot_table_entry(p, plan, st.index, 0, w, ci, 0, f) + pedslot_table_entry(p, plan, st.index, 1, wave, ci, 0, f) + pedslot_table_entry(p, plan, st.index, 2, lane, ci, t, f);
						for (uint32_t f = 0; f < 16; ++f) k[f] = p.fterms[(size_t)(run.c0 + ci) * PSLOT_FSTRIDE4 + 4 + f].c;
						cost = pslot_fact4_cost(L4[0], L4[1], L4[2], L4[3], k, bit(ex.pad[0], t), bit(ex.pad[0], 16 + t), bit(ex.pad[1], t), bit(ex.pad[1], 16 + t));
					} else
					for (uint32_t f = 0; f < ex.nf; ++f) {
						const uint32_t a = pedslot_table_entry(p, plan, st.index, 0, w, ci, t, f) + pedslot_table_entry(p, plan, st.index, 1, wave, ci, t, f);
						cost = std::min(cost, a + pedslot_table_entry(p, plan, st.index, 2, lane, ci, t, f));
					}
					const size_t me = (size_t)P * T + t;
					D[me] = sat_add_host(V[me], cost);
					records[st.index][(((size_t)w * ex.rec_words) + (size_t)(ci >> 2) * threads + tid) * 4u + (ci & 3u)] = (uint8_t)J[me];
				}
			}
			for (uint32_t q = 0; q < row.n_end; ++q) {
				const uint32_t slot = (q == 0 ? row.info0 : (q == 1 ? row.info1 : (q == 2 ? row.info2 : row.pad[2]))) & 255u;
				const uint32_t M = q == 0 ? row.M0 : (q == 1 ? row.M1 : (q == 2 ? row.M2 : row.pad[3]));
				if (slot >= L) { msg = "an ending read sits in a grid slot"; return false; }
				V = D;
				for (uint32_t P = 0; P < ncell; ++P) {
					const uint32_t qq = (uint32_t)__builtin_popcount(P & M) & 1u;
					const uint32_t w = P >> L, l = P & ((1u << L) - 1u);
					for (uint32_t t = 0; t < T; ++t) {
						const size_t me = (size_t)P * T + t;
						const uint32_t other = V[(size_t)(P ^ (1u << slot)) * T + t];
						if (other < sat_add_host(V[me], qq))
							records[st.index][(((size_t)w * ex.rec_words) + (size_t)(ci >> 2) * threads + ((l << TB) | t)) * 4u + (ci & 3u)] |= (uint8_t)(16u << q);
						D[me] = std::min(V[me], other);
					}
				}
			}
		}
		// exit
		const uint32_t out_size = run.out_fullmask + 1u;
		nx.assign((size_t)(out_size ? out_size : 1u) * T, 0xDEADBEEFu);
		const uint32_t localmask = (1u << L) - 1u;
		for (uint32_t P = 0; P < ncell; ++P) {
			if ((P & localmask) & ~run.out_occ) continue;   // representatives: free-slot bits zero
			uint32_t idx = 0;
			for (uint32_t s = 0; s < nslots; ++s) if (bit(run.out_occ, s)) idx |= bit(P, s) << slot_pos(run.out_pos, s);
			if ((size_t)idx * T + T > nx.size()) { msg = "pedigree run writes beyond the exchange buffer"; return false; }
			for (uint32_t t = 0; t < T; ++t) nx[(size_t)idx * T + t] = D[(size_t)P * T + t];
		}
		pr.swap(nx);
	}
	score = total;
	// ---- backtrace: state = (logical index at the first column of the unit walked before, transmission value handed down)
	uint32_t x = last_x, tprev = last_j;
	for (size_t si = plan.steps.size(); si-- > 0;) {
		const Step& st = plan.steps[si];
		if (st.kind == 0) {
			const uint32_t c = st.index;
			if (c + 1 == n) { path_index[c] = x; path_trans[c] = last_t; continue; }
			const uint32_t y = p.f[c] == 0 ? 0u : (x & ((1u << p.f[c]) - 1u));
			const uint32_t raw = col_arg[c][(size_t)y * T + tprev];
			path_index[c] = raw & 0x0FFFFFFFu;
			path_trans[c] = tprev;
			x = raw & 0x0FFFFFFFu;
			tprev = raw >> 28;
			continue;
		}
		const SlotRun& run = plan.runs[st.index];
		const PedSlotExtra& ex = plan.pextra[st.index];
		const uint32_t L = run.L, threads = run.threads, TB = ex.tb;
		const std::vector<uint8_t>& exs = plan.exit_slot[st.index];
		uint32_t pexit = 0;
		for (uint32_t j = 0; j < plan.f_exit[st.index]; ++j) pexit |= bit(x, j) << exs[j];
		const uint32_t w = pexit >> L;
		uint32_t l = pexit & ((1u << L) - 1u), tcur = tprev;
		auto rec_byte = [&](uint32_t ci, uint32_t cell, uint32_t t) {
			return (uint32_t)records[st.index][(((size_t)w * ex.rec_words) + (size_t)(ci >> 2) * threads + ((cell << TB) | t)) * 4u + (ci & 3u)];
		};
		for (uint32_t ci = run.ncols; ci-- > 0;) {
			const SlotBtCol& bc = plan.bt_cols[run.row_off + ci];
			for (uint32_t e = bc.pad[0]; e-- > 0;) {
				const uint32_t slot = e < 3u ? bc.slot[25 + e] : bc.pad[1];
				const uint32_t look = l & ~(1u << slot);
				l = look | (((rec_byte(ci, look, tcur) >> (4u + e)) & 1u) << slot);
			}
			const uint32_t pc = (w << L) | l;
			uint32_t xl = 0;
			for (uint32_t j = 0; j < bc.k; ++j) xl |= bit(pc, bc.slot[j]) << j;
			path_index[run.c0 + ci] = xl;
			path_trans[run.c0 + ci] = tcur;
			tcur = rec_byte(ci, l, tcur) & 15u;
		}
		x = path_index[run.c0];
		tprev = tcur;
	}
	return true;
}

}  // namespace whamd
