// slot_emulate.cpp -- HOST-ONLY DIAGNOSTIC of the slot-run planner (whamd_debug_emulate_slot_plan): executes a SlotPlan the
// way kernels_slots.h / kernels_backtrace.h do -- same physical indices, same per-slot deltas, same decision bits and
// record layout, same entry / exit layouts and mirror rules -- cell by cell on the CPU, so that the CPU test-suite can
// check the PLAN (slot assignment, tie-break masks, exchange layouts, backtrace blobs) against the oracle without a GPU.
// It is exponential-size bookkeeping for small inputs, not a solver: no product path calls it (the product path is
// the HIP library and fails loudly without a device).
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

// cost of cell x of column c from the host term list (T == 1)
uint32_t cell_cost(const Problem& p, uint32_t c, uint32_t x) {
	const int32_t* dl = p.delta.data() + (size_t)p.col_ptr[c];
	int64_t L = 0;
	for (uint32_t j = 0; j < p.k[c]; ++j) if (bit(x, j)) L += dl[j];
	uint32_t best = INF;
	for (uint64_t q = p.term_begin(c, 0); q < p.term_end(c, 0); ++q) {
		const CostTerm& t = p.terms[q];
		uint32_t v = t.c;
		if (t.plus) v += (uint32_t)L;
		if (t.minus) v -= (uint32_t)L;
		best = std::min(best, v);
	}
	return best;
}

}  // namespace

// Returns false (msg set) on an internal inconsistency.  path_index: [n_cols] logical bipartition index per column.
bool emulate_slot_plan(const Problem& p, const SlotPlan& plan, std::vector<uint32_t>& path_index, uint32_t& score, std::string& msg) {
	const uint32_t n = p.n_cols;
	path_index.assign(n, 0);
	score = 0;
	if (n == 0) return true;
	std::vector<uint32_t> pr, nx;                        // exchange buffers
	std::vector<std::vector<uint32_t>> col_arg(n);        // per-column steps: argmin cell per projection entry
	std::vector<std::vector<uint8_t>> records(plan.runs.size());
	uint32_t last_x = 0;
	uint32_t total = 0;
	for (size_t si = 0; si < plan.steps.size(); ++si) {
		const Step& st = plan.steps[si];
		if (st.kind == 0) {
			const uint32_t c = st.index, k = p.k[c], b = p.b[c], f = p.f[c];
			const bool is_last = c + 1 == n;
			nx.assign((size_t)1 << f, INF);
			col_arg[c].assign((size_t)1 << f, 0);
			std::vector<uint32_t> best_rank((size_t)1 << f, 0xFFFFFFFFu);
			uint32_t opt = INF, opt_rank = 0xFFFFFFFFu;
			for (uint32_t x = 0; x < (1u << k); ++x) {
				const uint32_t prev = c == 0 ? 0u : pr[x & ((1u << b) - 1u)];   // b == 0: the single value the previous component projected onto
				const uint32_t D = cell_cost(p, c, x) + prev;
				const uint32_t rank = gray_rank_host(x);
				if (is_last) {
					if (D < opt || (D == opt && rank < opt_rank)) { opt = D; opt_rank = rank; last_x = x; }
					continue;
				}
				uint32_t y = 0, fi = 0;
				for (uint32_t j = 0; j < k; ++j) if (bit(p.fwd_mask[c], j)) y |= bit(x, j) << fi++;
				if (D < nx[y] || (D == nx[y] && rank < best_rank[y])) { nx[y] = D; best_rank[y] = rank; col_arg[c][y] = x; }
			}
			if (is_last) total = opt;
			pr.swap(nx);
			continue;
		}
		const SlotRun& run = plan.runs[st.index];
		const uint32_t L = run.L, nslots = L + run.g, nwg = 1u << (run.g - run.half), threads = run.threads, LRr = run.lr, R = 1u << LRr;
		const uint32_t ncell = nwg << L;
		std::vector<uint32_t> D(ncell, 0), D2(ncell);
		records[st.index].assign((size_t)nwg * run.n_ends * threads, 0);
		if (run.has_prev) {
			for (uint32_t P = 0; P < ncell; ++P) {
				uint32_t idx = 0;
				if (run.in_identity) idx = P & run.in_occ;
				else for (uint32_t s = 0; s < nslots; ++s) if (bit(run.in_occ, s)) idx |= bit(P, s) << slot_pos(run.in_pos, s);
				if (run.in_half && bit(idx, run.in_mirror_pos)) idx ^= run.in_fullmask;
				if (idx >= pr.size()) { msg = "slot run reads beyond the exchange buffer"; return false; }
				D[P] = pr[idx];
			}
		}
		// Y form (slot_plan.cpp): the cells hold Y = B - 2 D; the row's first four words are Kr[0..3], the tables are doubled and biased
		const bool yf = run.yflags & 1u;
		if (yf && !(run.yflags & 2u)) for (uint32_t P = 0; P < ncell; ++P) D[P] = run.base_in - 2u * D[P];
		uint32_t k_end = 0;
		for (uint32_t ci = 0; ci < run.ncols; ++ci) {
			const SlotRow& row = plan.rows[run.row_off + ci];
			for (uint32_t P = 0; P < ncell; ++P) {
				uint32_t A = row.Cp;
				for (uint32_t s = LRr + SLOT_LANE; s < nslots; ++s) if (bit(P, s)) A += (uint32_t)row.dslot[s];
				for (uint32_t s = 0; s < (uint32_t)SLOT_LANE; ++s) if (bit(P, LRr + s)) A += (uint32_t)row.dslot[LRr + s];   // (dlane duplicates these; a Y-form row reuses its words)
				if (yf) {
					if (LRr != 2 && LRr != 3) { msg = "a Y-form run has two or three reg slots"; return false; }
					const uint32_t* kr = reinterpret_cast<const uint32_t*>(&row);   // Kr[0 .. 2^lr)
					const uint32_t x0 = 2u * A + SLOT_YBIAS, k = kr[P & (R - 1u)];   // (A of the thread's cell 0: reg-slot bits not added)
					D[P] += x0 > k ? x0 - k : k - x0;
					continue;
				}
				for (uint32_t s = 0; s < LRr; ++s) if (bit(P, s)) A += (uint32_t)row.dreg[s];
				D[P] += std::min(std::min(A, row.K - A), row.Cc);
			}
			for (uint32_t q = 0; q < row.n_end; ++q) {
				const uint32_t info = row.end[q].info, M = row.end[q].M;
				const uint32_t slot = info & 255u, qmask = (info >> 8) & 0xFFFFu, mflip = (info >> 24) & 1u;
				if (slot >= L) { msg = "an ending read sits in a grid slot"; return false; }
				for (uint32_t P = 0; P < ncell; ++P) {
					const uint32_t Pthr = P & ~(R - 1u), r = P & (R - 1u);
					uint32_t qthr = (uint32_t)__builtin_popcount(Pthr & M) & 1u;
					(void)mflip;   // (lane / wave slots: folded into M by the planner)
					const uint32_t qq = qthr ^ bit(qmask, r);
					const uint32_t other = D[P ^ (1u << slot)];
					const uint32_t w = P >> L, tid = (P & ((1u << L) - 1u)) >> LRr;
					const bool takes = yf ? (D[P] < other + qq) : (other < D[P] + qq);   // Y form: the larger Y is the smaller D
					if (takes) records[st.index][((size_t)w * run.n_ends + k_end) * threads + tid] |= (uint8_t)(1u << r);
					D2[P] = yf ? std::max(D[P], other) : std::min(D[P], other);
				}
				D.swap(D2);
				++k_end;
			}
		}
		// exit
		if (yf && !(run.yflags & 4u)) {
			for (uint32_t P = 0; P < ncell; ++P) {
				if ((run.base_out - D[P]) & 1u) { msg = "Y-form exit: B - Y is odd"; return false; }
				if (D[P] > run.base_out) { msg = "Y-form exit: Y exceeds its base"; return false; }
				D[P] = (run.base_out - D[P]) >> 1;
			}
		}
		const uint32_t out_size = run.out_fullmask + 1u;
		nx.assign(out_size ? out_size : 1u, 0xDEADBEEFu);
		const uint32_t localmask = (1u << L) - 1u;
		for (uint32_t P = 0; P < ncell; ++P) {
			if ((P & localmask) & ~run.out_occ) continue;   // representatives: free-slot bits zero
			uint32_t idx = 0;
			for (uint32_t s = 0; s < nslots; ++s) if (bit(run.out_occ, s)) idx |= bit(P, s) << slot_pos(run.out_pos, s);
			if (idx >= nx.size()) { msg = "slot run writes beyond the exchange buffer"; return false; }
			nx[idx] = D[P];
			if (run.mirror_out) nx[idx ^ run.out_fullmask] = D[P];
		}
		pr.swap(nx);
	}
	score = total;
	// (one job: the value a connected component projects onto is carried into the next component's cells, so the last
	// column's optimum is the total -- the device splits components into jobs and adds their scores on the host instead)
	// ---- backtrace (kernels_backtrace.h): newest step first
	uint32_t x = last_x;
	for (size_t si = plan.steps.size(); si-- > 0;) {
		const Step& st = plan.steps[si];
		if (st.kind == 0) {
			const uint32_t c = st.index;
			if (c + 1 == n) { path_index[c] = x; continue; }
			const uint32_t y = p.f[c] == 0 ? 0u : (x & ((1u << p.f[c]) - 1u));
			x = col_arg[c][y];
			path_index[c] = x;
			continue;
		}
		const SlotRun& run = plan.runs[st.index];
		const uint32_t L = run.L, threads = run.threads, LRr = run.lr, R = 1u << LRr;
		const std::vector<uint8_t>& ex = plan.exit_slot[st.index];
		uint32_t pexit = 0;
		for (uint32_t j = 0; j < plan.f_exit[st.index]; ++j) pexit |= bit(x, j) << ex[j];
		const uint32_t w = pexit >> L;
		uint32_t l = pexit & ((1u << L) - 1u);
		const bool mirrored = run.half && bit(w, run.g - 1u);
		const uint32_t wrec = mirrored ? (~w & ((1u << run.g) - 1u)) : w;
		if (wrec >= (1u << (run.g - run.half))) { msg = "backtrace reads the record of a workgroup that was not launched"; return false; }
		const uint32_t lmask = (1u << L) - 1u;
		std::vector<uint32_t> cells(run.n_ends + 1);
		cells[run.n_ends] = l;
		const uint8_t* ends = plan.end_slots.data() + plan.end_off[st.index];
		for (uint32_t k = run.n_ends; k-- > 0;) {
			const uint32_t j = ends[k];
			const uint32_t look = mirrored ? ((~l & lmask) | (1u << j)) : (l & ~(1u << j));
			const uint32_t byte = records[st.index][((size_t)wrec * run.n_ends + k) * threads + (look >> LRr)];
			const uint32_t b = (byte >> (look & (R - 1u))) & 1u;
			l = (l & ~(1u << j)) | (b << j);
			cells[k] = l;
		}
		for (uint32_t ci = 0; ci < run.ncols; ++ci) {
			const SlotBtCol& bc = plan.bt_cols[run.row_off + ci];
			const uint32_t pc = (w << L) | cells[bc.kf];
			uint32_t xl = 0;
			for (uint32_t j = 0; j < bc.k; ++j) xl |= bit(pc, bc.slot[j]) << j;
			path_index[run.c0 + ci] = xl;
		}
		x = path_index[run.c0];
	}
	return true;
}

}  // namespace whamd
