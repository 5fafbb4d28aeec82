// Round-6 host-side sanitizer pass (AddressSanitizer + UBSan, CPU only -- the GPU pool has neither): build_problem, plan_forward_slots, finish_solution on seeded random
// tables (one individual regular / ragged, homozygous columns, BLANK entries, a trio) from several threads at once.  The per-entry passes changed in round 6 (selects
// instead of branches, ending reads by set bits, the view's arrays read in place) run here under the sanitizers.
// Build (from whatshap_amd/csrc): g++ -std=c++17 -O1 -g -fsanitize=address,undefined -pthread -o /tmp/r6san ../../scripts/micro/r6_host_sanitize.cpp problem.cpp slot_plan.cpp resident_plan.cpp host_memory.cpp
#include <cstdio>
#include <random>
#include <thread>
#include <vector>
#include "../../whatshap_amd/csrc/problem.h"
#include "../../whatshap_amd/csrc/slots.h"
using namespace whamd;
struct Table {
	std::vector<uint64_t> read_ptr; std::vector<int32_t> pos; std::vector<uint8_t> al; std::vector<uint32_t> q; std::vector<int32_t> sid;
	std::vector<uint32_t> recomb, positions, ind, triples; std::vector<uint8_t> geno;
};
static Table make(uint32_t n, uint32_t cov, uint32_t seed, bool trio, bool ragged) {
	Table t; std::mt19937 rng(seed);
	t.read_ptr.push_back(0);
	const uint32_t n_ind = trio ? 3 : 1;
	for (uint64_t i = 0;; ++i) {
		const uint32_t len = ragged ? 2 + rng() % 40 : 30;
		const uint64_t start = i * 30 / cov;
		if (start + 2 > n) break;
		size_t before = t.pos.size();
		for (uint32_t c = (uint32_t)start; c < std::min<uint64_t>(n, start + len); ++c) {
			if (ragged && rng() % 7 == 0 && t.pos.size() > before) continue;   // a hole: the flattener inserts BLANK
			t.pos.push_back((int32_t)(c * 10 + 1)); t.al.push_back(rng() % 11 == 0 ? 2 : (rng() & 1)); t.q.push_back(rng() % 5 == 0 ? 0 : 5 + rng() % 30);
		}
		if (t.pos.size() - before < 2) { t.pos.resize(before); t.al.resize(before); t.q.resize(before); continue; }
		t.read_ptr.push_back(t.pos.size()); t.sid.push_back(trio ? (int32_t)(7 + i % 3) : 7);
	}
	t.positions.resize(n); for (uint32_t c = 0; c < n; ++c) t.positions[c] = c * 10 + 1;
	t.recomb.assign(n, 10);
	t.geno.resize((size_t)n_ind * n);
	for (auto& g : t.geno) g = trio ? 1 : (rng() % 9 == 0 ? (rng() & 1) * 2 : 1);
	if (trio) { t.ind = {7, 8, 9}; t.triples = {7, 8, 9}; for (uint32_t c = 0; c < n; ++c) { t.geno[c] = 1; t.geno[n + c] = 1; t.geno[2 * (size_t)n + c] = rng() % 2 ? 1 : (rng() & 1) * 2; } }
	else t.ind = {7};
	return t;
}
int main() {
	std::vector<std::thread> th;
	for (int w = 0; w < 6; ++w) th.emplace_back([w] {
		for (int rep = 0; rep < 6; ++rep) {
			const bool trio = (w % 3) == 2, ragged = (w % 2) == 1;
			const Table t = make(1500 + 700 * rep, trio ? 4 + rep % 3 : 6 + 2 * rep, 1000 * w + rep, trio, ragged);
			whamd_readset_view rs{(uint32_t)t.sid.size(), t.read_ptr.data(), t.pos.data(), t.al.data(), t.q.data(), t.sid.data()};
			whamd_pedigree_view pv{(uint32_t)t.ind.size(), t.ind.data(), (uint32_t)(t.triples.size() / 3), t.triples.empty() ? nullptr : t.triples.data(), (uint32_t)t.positions.size(), t.geno.data(), nullptr, nullptr};
			Problem p; std::string msg;
			const whamd_status_t st = build_problem(&rs, t.recomb.data(), t.recomb.size(), &pv, false, t.positions.data(), t.positions.size(), p, msg);
			if (st != WHAMD_OK) { printf("thread %d rep %d: build_problem: %s\n", w, rep, msg.c_str()); continue; }
			SlotPlan plan;
			const bool planned = plan_forward_slots(p, 11, 1, plan, 2 + rep % 2, false);
			Solution s;
			s.path_index.assign(p.n_cols, 0); s.path_trans.assign(p.n_cols, 0);
			std::mt19937 rng(w * 77 + rep);
			for (uint32_t c = 0; c < p.n_cols; ++c) { s.path_index[c] = rng() & ((1u << p.k[c]) - 1u); s.path_trans[c] = rng() % p.T; }   // (any path: the loops must stay inside their arrays)
			std::string m2;
			const whamd_status_t fs = finish_solution(p, s, m2);
			printf("thread %d rep %d: %u columns, max coverage %u, %s, plan %s (%zu runs), finish %d\n", w, rep, p.n_cols, p.max_k, trio ? "trio" : "single", planned ? "ok" : "declined", plan.runs.size(), (int)fs);
		}
	});
	for (auto& x : th) x.join();
	return 0;
}
