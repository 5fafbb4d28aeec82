// c_api.cpp -- the extern "C" boundary declared in include/whatshap_amd.h.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>

#include "debug_build.h"
#ifdef WHAMD_DEBUG_BUILD
#include "../../include/whatshap_amd_debug.h"
#endif
#include "device_table.h"
#include "genotype.h"
#include "heuristic.h"
#include "host_parallel.h"
#include "problem.h"
#include "resident.h"
#include "slots.h"
#include <algorithm>
#include <vector>

using namespace whamd;

struct whamd_dptable {
	Problem problem;
	Solution solution;
	DeviceTable device;
	whamd_solve_stats stats{};
	int device_index = 0;
	bool uploaded = false;
	bool solved = false;
	bool in_flight = false;
};

namespace {

thread_local std::string g_last_error;

whamd_status_t fail(whamd_status_t st, const std::string& msg) {
	g_last_error = msg;
	return st;
}

// No C++ exception crosses the C boundary: std::bad_alloc of the flatten / plan vectors, std::system_error of a worker thread that could not
// be started, anything a worker carried over (host_parallel.h) become WHAMD_ERR_HOST with the exception's message.
template <class F>
whamd_status_t guarded(F&& body) {
	try {
		return body();
	} catch (const std::bad_alloc&) {
		return fail(WHAMD_ERR_HOST, "out of host memory");
	} catch (const std::exception& e) {
		return fail(WHAMD_ERR_HOST, std::string("host-side failure: ") + e.what());
	} catch (...) {
		return fail(WHAMD_ERR_HOST, "host-side failure (unknown exception)");
	}
}

double now_ms() {
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

namespace whamd {
// for the host-only entry points that live in other files (readselect.cpp)
void set_last_error(const std::string& msg) { g_last_error = msg; }
}

extern "C" {

int whamd_abi_version(void) { return WHAMD_ABI_VERSION; }

int whamd_device_count(void) { return DeviceTable::device_count(); }

whamd_status_t whamd_device_pci_bus_id(int device, char* out, size_t capacity) {
	if (!out || capacity < 16) return fail(WHAMD_ERR_INVALID, "whamd_device_pci_bus_id: buffer of at least 16 bytes expected");
	std::string id;
	if (!DeviceTable::device_pci_bus_id(device, id)) return fail(WHAMD_ERR_DEVICE, "no HIP device " + std::to_string(device));
	if (id.size() + 1 > capacity) return fail(WHAMD_ERR_INVALID, "whamd_device_pci_bus_id: buffer too small");
	std::memcpy(out, id.c_str(), id.size() + 1);
	return WHAMD_OK;
}

const char* whamd_last_error(void) { return g_last_error.c_str(); }

namespace {
whamd_status_t apply_option(whamd_dptable* t, const std::string& k, const char* value);   // (below: whamd_dptable_set_option)

whamd_status_t create_table(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost, const whamd_pedigree_view* pedigree,
                            int distrust_genotypes, const uint32_t* positions, size_t n_positions, const char* const* keys, const char* const* values,
                            size_t n_options, int device, whamd_dptable** out) {
	if (!out) return fail(WHAMD_ERR_INVALID, "out is NULL");
	*out = nullptr;
	if (n_options && (!keys || !values)) return fail(WHAMD_ERR_INVALID, "option arrays are NULL");
	const double t0 = now_ms();
	std::unique_ptr<whamd_dptable> t(new whamd_dptable());
	std::string msg;
	// "host_threads": how many host threads THIS create may use (flatten, plan, staging copies); restored on every way out
	struct ThreadsGuard { uint32_t saved = whamd::host_threads_override(); ~ThreadsGuard() { whamd::host_threads_override() = saved; } } threads_guard;
	for (size_t i = 0; i < n_options; ++i) {
		if (!keys[i] || !values[i]) return fail(WHAMD_ERR_INVALID, "null option");
		if (std::string(keys[i]) == "host_threads") whamd::host_threads_override() = (uint32_t)std::max(0, std::atoi(values[i]));
	}
	whamd_status_t st = build_problem(readset, recombcost, n_recombcost, pedigree, distrust_genotypes != 0,
	                                  positions, n_positions, t->problem, msg, /*columns_only=*/false, /*lazy_fact_terms=*/true);
	if (st != WHAMD_OK) return fail(st, msg);
	const double t1 = now_ms();
	t->device_index = device;
	for (size_t i = 0; i < n_options; ++i) {   // before the (one) upload: the plan is made for them
		if (std::string(keys[i]) == "host_threads") continue;
		st = apply_option(t.get(), keys[i], values[i]);
		if (st != WHAMD_OK) return st;
	}
	st = t->device.upload(t->problem, device, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	if (getenv("WHAMD_DEBUG_TIMING"))
		fprintf(stderr, "[whamd timing] create: flatten %.1f ms, plan + upload %.1f ms\n", t1 - t0, now_ms() - t1);
	t->uploaded = true;
	t->stats.host_prepare_ms = now_ms() - t0;
	t->stats.host_flatten_ms = t1 - t0;
	*out = t.release();
	return WHAMD_OK;
}
}  // namespace

whamd_status_t whamd_dptable_create(const whamd_readset_view* readset, const uint32_t* recombcost,
                                    size_t n_recombcost, const whamd_pedigree_view* pedigree,
                                    int distrust_genotypes, const uint32_t* positions, size_t n_positions,
                                    int device, whamd_dptable** out) {
	return guarded([&]() -> whamd_status_t {
		return create_table(readset, recombcost, n_recombcost, pedigree, distrust_genotypes, positions, n_positions, nullptr, nullptr, 0, device, out);
	});
}

whamd_status_t whamd_dptable_create_with_options(const whamd_readset_view* readset, const uint32_t* recombcost,
                                                 size_t n_recombcost, const whamd_pedigree_view* pedigree,
                                                 int distrust_genotypes, const uint32_t* positions, size_t n_positions,
                                                 const char* const* keys, const char* const* values, size_t n_options,
                                                 int device, whamd_dptable** out) {
	return guarded([&]() -> whamd_status_t {
		return create_table(readset, recombcost, n_recombcost, pedigree, distrust_genotypes, positions, n_positions, keys, values, n_options, device, out);
	});
}

namespace {

// Everything whamd_dptable_enqueue does before the first launch.
whamd_status_t begin_enqueue(whamd_dptable* t) {
	if (!t) return fail(WHAMD_ERR_INVALID, "table is NULL");
	if (t->in_flight) return fail(WHAMD_ERR_INVALID, "a solve of this table is already in flight");
	std::string msg;
	if (!t->uploaded) {
		whamd_status_t st = t->device.upload(t->problem, t->device_index, msg);
		if (st != WHAMD_OK) return fail(st, msg);
		t->uploaded = true;
	}
	const Problem& p = t->problem;
	whamd_solve_stats& s = t->stats;
	s.n_columns = p.n_cols;
	s.n_cells = p.n_cells;
	s.n_costs = p.n_cells * p.T;
	s.algorithmic_bytes = p.algorithmic_bytes;
	s.max_coverage = p.max_k;
	s.transmissions = p.T;
	t->solved = false;
	return WHAMD_OK;
}

}  // namespace

whamd_status_t whamd_dptable_enqueue(whamd_dptable* t) {
	return guarded([&]() -> whamd_status_t {
	whamd_status_t st = begin_enqueue(t);
	if (st != WHAMD_OK) return st;
	std::string msg;
	t->device.set_side_by_side(false);   // (a table submitted by itself has the device to itself as far as the library knows: whamd_dptable_enqueue_many says otherwise)
	st = t->device.enqueue(t->problem, t->solution, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	t->in_flight = true;
	return WHAMD_OK;
	});
}

whamd_status_t whamd_dptable_enqueue_many(whamd_dptable* const* tables, size_t n_tables) {
	return guarded([&]() -> whamd_status_t {
	if (!tables && n_tables) return fail(WHAMD_ERR_INVALID, "tables is NULL");
	{
		std::vector<const whamd_dptable*> seen(tables, tables + n_tables);
		std::sort(seen.begin(), seen.end());
		if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) return fail(WHAMD_ERR_INVALID, "a table appears twice in the list");
	}
	for (size_t i = 0; i < n_tables; ++i) {
		whamd_status_t st = begin_enqueue(tables[i]);
		if (st != WHAMD_OK) return st;
	}
	std::string msg;
	// Tables on slot runs (the default path) of one device advance TOGETHER: one launch per super-step serves all of them
	// (DeviceTable::enqueue_group).  Everything else -- other paths, windowed tables, a device with a single table -- keeps its own
	// stream; those submissions are interleaved round robin below.
	std::vector<size_t> rest;
	{
		std::vector<std::pair<int, size_t>> eligible;   // (device, position)
		for (size_t i = 0; i < n_tables; ++i) {
			if (tables[i]->device.group_eligible(tables[i]->problem)) eligible.emplace_back(tables[i]->device_index, i);
			else rest.push_back(i);
		}
		std::stable_sort(eligible.begin(), eligible.end(), [](const std::pair<int, size_t>& a, const std::pair<int, size_t>& b) { return a.first < b.first; });
		for (size_t a = 0; a < eligible.size();) {
			size_t b = a;
			while (b < eligible.size() && eligible[b].first == eligible[a].first) ++b;
			// up to four tables that fill the chip by themselves (>= 128 workgroups per launch) do better on their own streams: their
			// launches drift against each other, one table's boundary and prologue under another's columns (5.4 M columns/s for three
			// coverage-20 tables against 3.5 M in lockstep); many tables, or narrow ones, share their launches
			bool wide = true;
			for (size_t q = a; q < b; ++q) wide = wide && tables[eligible[q].second]->device.widest_launch() >= 128;
			if (b - a == 1 || (b - a <= 4 && wide && !getenv("WHAMD_GROUP_ALWAYS"))) {
				for (size_t q = a; q < b; ++q) rest.push_back(eligible[q].second);
				a = b;
				continue;
			}
			std::vector<DeviceTable*> devs;
			std::vector<const Problem*> probs;
			std::vector<Solution*> sols;
			for (size_t q = a; q < b; ++q) {
				whamd_dptable* t = tables[eligible[q].second];
				devs.push_back(&t->device); probs.push_back(&t->problem); sols.push_back(&t->solution);
			}
			const whamd_status_t st = DeviceTable::enqueue_group(devs.data(), probs.data(), sols.data(), devs.size(), msg);
			if (st != WHAMD_OK) return fail(st, msg);   // (the group has rewound itself; tables of earlier groups stay in flight)
			for (size_t q = a; q < b; ++q) tables[eligible[q].second]->in_flight = true;
			a = b;
		}
		std::sort(rest.begin(), rest.end());
	}
	// round robin over the remaining tables, a few launches each: their streams fill up side by side
	for (size_t i : rest) tables[i]->device.set_side_by_side(rest.size() > 1);
	constexpr uint64_t SLICE = 16;
	std::vector<uint8_t> done(n_tables, 1);
	for (size_t i : rest) done[i] = 0;
	size_t open = rest.size();
	while (open) {
		for (size_t i : rest) {
			if (done[i]) continue;
			bool finished = false;
			whamd_status_t st = tables[i]->device.enqueue_some(tables[i]->problem, tables[i]->solution, SLICE, finished, msg);
			if (st != WHAMD_OK) {
				// the failing table has rewound itself; the others must not keep half-submitted schedules either: tables
				// whose submission is complete stay in flight (collect them with whamd_dptable_wait), the rest is aborted
				for (size_t j : rest)
					if (j != i && !done[j]) tables[j]->device.abort_enqueue();
				return fail(st, msg);
			}
			if (finished) {
				done[i] = 1;
				tables[i]->in_flight = true;
				--open;
			}
		}
	}
	return WHAMD_OK;
	});
}

whamd_status_t whamd_dptable_wait(whamd_dptable* t) {
	return guarded([&]() -> whamd_status_t {
	if (!t) return fail(WHAMD_ERR_INVALID, "table is NULL");
	if (!t->in_flight) return fail(WHAMD_ERR_INVALID, "whamd_dptable_enqueue has not run");
	t->in_flight = false;
	std::string msg;
	whamd_status_t st = t->device.wait(t->problem, t->solution, t->stats, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	const double t0 = now_ms();
	st = finish_solution(t->problem, t->solution, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	t->stats.host_finish_ms = now_ms() - t0;
	t->solved = true;
	return WHAMD_OK;
	});
}

whamd_status_t whamd_dptable_wait_many(whamd_dptable* const* tables, size_t n_tables) {
	return guarded([&]() -> whamd_status_t {
	if (!tables && n_tables) return fail(WHAMD_ERR_INVALID, "tables is NULL");
	for (size_t i = 0; i < n_tables; ++i) {
		if (!tables[i]) return fail(WHAMD_ERR_INVALID, "table is NULL");
		if (!tables[i]->in_flight) return fail(WHAMD_ERR_INVALID, "whamd_dptable_enqueue has not run");
	}
	{   // a table listed twice would be finished by two host threads at once
		std::vector<const whamd_dptable*> seen(tables, tables + n_tables);
		std::sort(seen.begin(), seen.end());
		if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) return fail(WHAMD_ERR_INVALID, "a table appears twice in the list");
	}
	// Every table's device side (its stream drained: path and scores have arrived in the pinned buffer) and, right behind it, its host side (superreads,
	// partitioning: get_super_reads / get_optimal_partitioning of the reference) -- on several host threads, a few tables each.  (Round 4 waited for the
	// tables one after the other on the calling thread and only then finished them in parallel: 96 tables spent 30 ms in that first loop.)
	whamd_status_t first = WHAMD_OK;
	std::string first_msg;
	std::vector<whamd_status_t> status(n_tables, WHAMD_OK);
	std::vector<std::string> messages(n_tables);
	for (size_t i = 0; i < n_tables; ++i) tables[i]->in_flight = false;
	// (at most 32 workers: one per table up to the CPUs of the host was tried -- 96 threads inside hipStreamSynchronize at once made a 96-table step 178 ms
	// instead of 98, the runtime's waiters contend)
	const uint32_t outer = (uint32_t)std::min<uint64_t>(host_threads(n_tables, 1), n_tables);   // (host_threads() returns n / grain + 1: never more workers than tables)
	const uint32_t inner = std::max(1u, host_threads(1u << 30, 1) / outer);   // (a table's own finish splits its columns over threads: not 32 x 7 of them at once)
	auto finish_one = [&](size_t i) {
		whamd_dptable* t = tables[i];
		const double t0 = now_ms();
		status[i] = finish_solution(t->problem, t->solution, messages[i]);
		t->stats.host_finish_ms = now_ms() - t0;
		t->solved = status[i] == WHAMD_OK;
	};
	// (A second phase -- the device waits on the 32 workers first, then every table's host side on a worker of its own -- was measured on one box, same process,
	// twelve steps each: 54.4 ms against 50.7 ms without; not kept.  Since then the host side of a single-individual table is 0.3 ms: nothing to spread.)
	const bool timing = getenv("WHAMD_DEBUG_TIMING") != nullptr;
	const double t_wait0 = now_ms();
	{
		std::vector<DeviceTable*> devs(n_tables);
		for (size_t i = 0; i < n_tables; ++i) devs[i] = &tables[i]->device;
		DeviceTable::wait_last_of_each_stream(devs.data(), n_tables);
	}
	std::vector<double> t_begun(timing ? n_tables : 0, 0.0), t_synced(timing ? n_tables : 0, 0.0), t_finished(timing ? n_tables : 0, 0.0);
	parallel_ranges(n_tables, outer, [&](uint64_t i0, uint64_t i1, uint32_t) {
		struct Budget { uint32_t saved = whamd::host_threads_override(); ~Budget() { whamd::host_threads_override() = saved; } } budget;
		whamd::host_threads_override() = inner;
		for (uint64_t i = i0; i < i1; ++i) {
			whamd_dptable* t = tables[i];
			if (timing) t_begun[i] = now_ms() - t_wait0;
			status[i] = t->device.wait(t->problem, t->solution, t->stats, messages[i]);
			if (timing) t_synced[i] = now_ms() - t_wait0;
			if (status[i] != WHAMD_OK) continue;
			finish_one(i);
			if (timing) t_finished[i] = now_ms() - t_wait0;
		}
	});
	if (timing && n_tables > 1) {
		double s0 = 1e30, s1 = 0, f1 = 0;
		for (size_t i = 0; i < n_tables; ++i) { s0 = std::min(s0, t_synced[i]); s1 = std::max(s1, t_synced[i]); f1 = std::max(f1, t_finished[i]); }
		double fin = 0, late_wait = 0, late_max = 0; size_t late = 0;
		for (size_t i = 0; i < n_tables; ++i) {
			fin += t_finished[i] - t_synced[i];
			if (t_begun[i] > s0) { late_wait += t_synced[i] - t_begun[i]; late_max = std::max(late_max, t_synced[i] - t_begun[i]); ++late; }   // (a wait that began after the first one had returned: the device was done)
		}
		fprintf(stderr, "[whamd timing] wait_many of %zu tables on %u workers: first table's device side done after %.1f ms, last after %.1f ms, last host side after %.1f ms; host side %.2f ms per table; %zu waits begun after the device was done took %.2f ms each (longest %.2f)\n",
		        n_tables, outer, s0, s1, f1, fin / n_tables, late, late ? late_wait / late : 0.0, late_max);
	}
	for (size_t i = 0; i < n_tables && first == WHAMD_OK; ++i)
		if (status[i] != WHAMD_OK) { first = status[i]; first_msg = messages[i]; }
	return first == WHAMD_OK ? WHAMD_OK : fail(first, first_msg);
	});
}

whamd_status_t whamd_dptable_solve(whamd_dptable* t) {
	return guarded([&]() -> whamd_status_t {
	whamd_status_t st = whamd_dptable_enqueue(t);
	if (st != WHAMD_OK) return st;
	return whamd_dptable_wait(t);
	});
}

whamd_status_t whamd_dptable_release_device(whamd_dptable* t) {
	if (!t) return fail(WHAMD_ERR_INVALID, "table is NULL");
	if (t->in_flight) return fail(WHAMD_ERR_INVALID, "a solve is in flight: call whamd_dptable_wait first");
	t->device.release_device();
	t->uploaded = false;
	return WHAMD_OK;
}

void whamd_dptable_destroy(whamd_dptable* t) {
	if (!t) return;
	if (getenv("WHAMD_DEBUG_TIMING")) {
		const double t0 = now_ms();
		t->device.release_device();
		const double t1 = now_ms();
		delete t;
		fprintf(stderr, "[whamd timing] destroy: device side %.2f ms, host side %.2f ms\n", t1 - t0, now_ms() - t1);
		return;
	}
	delete t;
}

uint64_t whamd_dptable_column_count(const whamd_dptable* t) { return t ? t->problem.n_cols : 0; }
uint32_t whamd_dptable_individual_count(const whamd_dptable* t) { return t ? t->problem.n_ind : 0; }
uint32_t whamd_dptable_read_count(const whamd_dptable* t) { return t ? t->problem.n_reads : 0; }

whamd_status_t whamd_dptable_positions(const whamd_dptable* t, uint32_t* out) {
	if (!t || !out) return fail(WHAMD_ERR_INVALID, "null argument");
	std::memcpy(out, t->problem.positions.data(), t->problem.positions.size() * sizeof(uint32_t));
	return WHAMD_OK;
}

#define REQUIRE_SOLVED(t)                                                            \
	if (!(t)) return fail(WHAMD_ERR_INVALID, "table is NULL");                       \
	if (!(t)->solved) return fail(WHAMD_ERR_INVALID, "whamd_dptable_solve has not run")

whamd_status_t whamd_dptable_get_optimal_score(const whamd_dptable* t, uint32_t* score_out) {
	REQUIRE_SOLVED(t);
	*score_out = t->solution.optimal_score;
	return WHAMD_OK;
}

whamd_status_t whamd_dptable_get_super_reads(const whamd_dptable* t, uint8_t* allele0_out, uint8_t* allele1_out,
                                             uint32_t* quality_out, uint32_t* transmission_out,
                                             uint32_t* sample_id_out) {
	REQUIRE_SOLVED(t);
	const Solution& s = t->solution;
	if (allele0_out) std::memcpy(allele0_out, s.allele0.data(), s.allele0.size());
	if (allele1_out) std::memcpy(allele1_out, s.allele1.data(), s.allele1.size());
	if (quality_out) std::memcpy(quality_out, s.quality.data(), s.quality.size() * sizeof(uint32_t));
	if (transmission_out) std::memcpy(transmission_out, s.path_trans.data(), s.path_trans.size() * sizeof(uint32_t));
	if (sample_id_out) std::memcpy(sample_id_out, t->problem.individual_id.data(), t->problem.individual_id.size() * sizeof(uint32_t));
	return WHAMD_OK;
}

whamd_status_t whamd_dptable_get_optimal_partitioning(const whamd_dptable* t, uint8_t* partition_out) {
	REQUIRE_SOLVED(t);
	std::memcpy(partition_out, t->solution.partition.data(), t->solution.partition.size());
	return WHAMD_OK;
}

whamd_status_t whamd_dptable_get_index_path(const whamd_dptable* t, uint32_t* index_out, uint32_t* transmission_out) {
	REQUIRE_SOLVED(t);
	const Solution& s = t->solution;
	if (index_out) std::memcpy(index_out, s.path_index.data(), s.path_index.size() * sizeof(uint32_t));
	if (transmission_out) std::memcpy(transmission_out, s.path_trans.data(), s.path_trans.size() * sizeof(uint32_t));
	return WHAMD_OK;
}

whamd_status_t whamd_dptable_get_stats(const whamd_dptable* t, whamd_solve_stats* stats_out) {
	if (!t || !stats_out) return fail(WHAMD_ERR_INVALID, "null argument");
	whamd_dptable* tt = const_cast<whamd_dptable*>(t);   // (the timings are a cache filled at the first request)
	if (t->solved && !t->in_flight) tt->device.read_timing(tt->stats);
	*stats_out = t->stats;
	return WHAMD_OK;
}

namespace {
whamd_status_t apply_option(whamd_dptable* t, const std::string& k, const char* value) {
	const std::string v(value);
	if (k == "path") {
		if (!t->device.set_path(v)) return fail(WHAMD_ERR_INVALID, "unknown path '" + v + "' (auto, slots, resident, column, column_keys)");
		t->uploaded = false;  // descriptors are rebuilt at the next solve
		return WHAMD_OK;
	}
	if (k == "resident_fold") {
		t->device.set_fold(v != "0");
		t->uploaded = false;
		return WHAMD_OK;
	}
	if (k == "symmetry") {
		t->device.set_symmetry(std::atoi(value));
		t->uploaded = false;
		return WHAMD_OK;
	}
	if (k == "lanes") {
		t->device.set_lanes(std::atoi(value));
		t->uploaded = false;
		return WHAMD_OK;
	}
	if (k == "resident_l") {
		t->device.set_l_pref(std::atoi(value));
		t->uploaded = false;
		return WHAMD_OK;
	}
	if (k == "slot_l") {
		t->device.set_slot_l(std::atoi(value));
		t->uploaded = false;
		return WHAMD_OK;
	}
	if (k == "slot_r") {
		t->device.set_slot_lr(std::atoi(value));
		t->uploaded = false;
		return WHAMD_OK;
	}
	if (k == "shared_launches") {
		t->device.set_shared_launches(v != "0");
		t->uploaded = false;
		return WHAMD_OK;
	}
	if (k == "arena_limit_bytes") {
		t->device.set_arena_limit(std::strtoull(value, nullptr, 10));
		t->uploaded = false;
		return WHAMD_OK;
	}
	return fail(WHAMD_ERR_INVALID, "unknown option '" + k + "'");
}
}  // namespace

whamd_status_t whamd_dptable_set_option(whamd_dptable* t, const char* key, const char* value) {
	if (!t || !key || !value) return fail(WHAMD_ERR_INVALID, "null argument");
	return apply_option(t, key, value);
}

whamd_status_t whamd_plan_summarize(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                    const whamd_pedigree_view* pedigree, int distrust_genotypes,
                                    const uint32_t* positions, size_t n_positions, const char* path,
                                    whamd_plan_summary* out) {
	return guarded([&]() -> whamd_status_t {
	if (!out) return fail(WHAMD_ERR_INVALID, "out is NULL");
	Problem p;
	std::string msg;
	const double tb0 = now_ms();
	const std::string mode(path ? path : "auto");
	whamd_status_t st = build_problem(readset, recombcost, n_recombcost, pedigree, distrust_genotypes != 0, positions,
	                                  n_positions, p, msg, /*columns_only=*/mode == "genotype_slots");
	if (st != WHAMD_OK) return fail(st, msg);
	SlotPlan sp;
	if (mode == "genotype_slots") {
		// the run plan of the genotyping path (genotype_slots.hip): every column must lie in a run, or the per-column kernels take over
		whamd_plan_summary s{};
		s.n_columns = p.n_cols;
		s.max_coverage = p.max_k;
		if (plan_forward_slots(p, 0, 0, sp, 0, /*genotype_mode=*/true)) {
			s.n_steps = sp.steps.size();
			s.n_runs = sp.runs.size();
			bool ok = true;
			uint32_t expect = 0;
			for (const Step& step : sp.steps) {
				if (step.kind != 2) {
					if (debug_env("WHAMD_DEBUG_PLAN")) fprintf(stderr, "[plan] genotype: column %u outside runs (k=%u b=%u f=%u, next k=%u)\n", step.index, p.k[step.index], p.b[step.index], p.f[step.index], step.index + 1 < p.n_cols ? p.k[step.index + 1] : 0);
					ok = ok && step.index == expect; expect = step.index + 1; continue;
				}
				const SlotRun& run = sp.runs[step.index];
				ok = ok && run.c0 == expect && run.lr == 0 && !run.half && run.ncols >= 1 && run.ncols <= (uint32_t)PSLOT_MAXCOLS;
				expect = run.c0 + run.ncols;
				s.n_resident_columns += run.ncols;
				s.max_run_columns = std::max<uint64_t>(s.max_run_columns, run.ncols);
				s.max_workgroups = std::max<uint64_t>(s.max_workgroups, 1ull << run.g);
				uint32_t starts = 0, ends = 0;
				for (uint32_t i = 0; i < run.ncols; ++i) {
					const PedSlotRow& pr = sp.prows[run.c0 + i];
					const uint32_t c = run.c0 + i;
					ok = ok && pr.pad[0] == (uint32_t)p.k[c] - (i == 0 ? p.b[run.c0] : p.b[c]) && pr.pad[1] == starts;
					ok = ok && pr.n_end == (c + 1 == p.n_cols ? 0u : (uint32_t)p.k[c] - p.f[c]) && sp.bt_cols[c].kf == ends;
					starts += pr.pad[0];
					ends += pr.n_end;
				}
			}
			ok = ok && expect == p.n_cols;
			s.invariants_ok = ok ? 1 : 0;
		}
		*out = s;
		return WHAMD_OK;
	}
	const double tp0 = now_ms();
	const bool slots_ok = (mode == "auto" || mode == "slots") && plan_forward_slots(p, 11, 1, sp);
	if (getenv("WHAMD_DEBUG_TIMING")) fprintf(stderr, "[whamd timing] plan_summarize: build_problem %.1f ms, plan_forward_slots %.1f ms\n", tp0 - tb0, now_ms() - tp0);
	if (slots_ok) {
		// slot runs (slots.h): every column in exactly one step, runs within their limits, slots consistent
		whamd_plan_summary s{};
		s.n_columns = p.n_cols;
		s.n_steps = sp.steps.size();
		s.n_runs = sp.runs.size();
		s.max_coverage = p.max_k;
		s.n_components = sp.component_first_step.size();
		bool ok = true;
		uint32_t expect = 0;
		size_t kc = 0;
		for (size_t si = 0; si < sp.steps.size(); ++si) {
			const Step& step = sp.steps[si];
			const uint32_t c0 = step.kind == 2 ? sp.runs[step.index].c0 : step.index;
			if (!sp.ped && (si == 0 || p.b[c0] == 0)) { ok = ok && kc < sp.component_first_step.size() && sp.component_first_step[kc] == si; ++kc; }
			ok = ok && c0 == expect;
			if (step.kind == 0) { expect = c0 + 1; ok = ok && sp.col_to_row[c0] < 0; continue; }
			const SlotRun& run = sp.runs[step.index];
			expect = c0 + run.ncols;
			ok = ok && step.kind == 2 && run.ncols >= 2 && run.ncols <= (uint32_t)(sp.ped ? PSLOT_MAXCOLS : SLOT_MAXCOLS) && run.g <= (uint32_t)SLOT_GMAX;
			if (sp.ped) {
				const PedSlotExtra& ex = sp.pextra[step.index];
				ok = ok && sp.pextra.size() == sp.runs.size() && run.lr == 0 && !run.half && (1u << ex.tb) == p.T && run.L == 6u - ex.tb + run.lw;
				ok = ok && (ex.nf == 2 || ex.nf == 4 || ((ex.nf == 16 || ex.nf == (uint32_t)PSLOT_FACT) && ex.tb == 2) || (ex.nf == (uint32_t)PSLOT_FACT4 && ex.tb == 4)) && ex.fwn == run.ncols * pslot_ta(ex.nf, p.T) * pslot_na(ex.nf) && ex.fwn <= (uint32_t)PSLOT_FORMWORDS && ex.rec_words == ((run.ncols + 3) / 4) * run.threads;
				ok = ok && run.lw <= (uint32_t)SLOT_LWMAX && run.threads == (64u << run.lw);
			} else
			ok = ok && run.lr >= 1 && run.lr <= (uint32_t)SLOT_LR && run.L == run.lr + (uint32_t)SLOT_LANE + run.lw && run.lw <= (uint32_t)SLOT_LWMAX && run.threads == (64u << run.lw);
			ok = ok && run.L + run.g <= (uint32_t)SLOT_MAXSLOTS && run.n_ends <= (uint32_t)SLOT_MAXENDS_RUN && (!run.half || run.g >= 1);
			uint32_t ends = 0;
			for (uint32_t i = 0; i < run.ncols && ok; ++i) {
				const uint32_t c = c0 + i;
				if (c + 1 >= p.n_cols) { ok = false; break; }   // the last column never runs inside a run
				ok = ok && sp.col_to_row[c] == (int32_t)(run.row_off + i) && (i == 0 || p.b[c] != 0);
				const uint32_t row_n_end = sp.ped ? sp.prows[run.row_off + i].n_end : sp.rows[run.row_off + i].n_end;
				const SlotBtCol& bc = sp.bt_cols[run.row_off + i];
				ok = ok && bc.k == p.k[c] && bc.kf == ends && row_n_end == (uint32_t)p.k[c] - p.f[c] && row_n_end <= (uint32_t)(sp.ped ? PSLOT_MAXEND : SLOT_MAXEND);
				if (sp.ped && sp.pextra[step.index].nf == (uint32_t)PSLOT_FACT) ok = ok && p.fterm_kind == 1 && p.fterms.size() == (size_t)p.n_cols * p.T * 16;
				else if (sp.ped && sp.pextra[step.index].nf == (uint32_t)PSLOT_FACT4) ok = ok && p.fterm_kind == 2 && p.fterms.size() == (size_t)p.n_cols * PSLOT_FSTRIDE4;
				else if (sp.ped) for (uint32_t t = 0; t < p.T; ++t) ok = ok && p.term_end(c, t) - p.term_begin(c, t) <= sp.pextra[step.index].nf;
				uint32_t used = 0;
				for (uint32_t j = 0; j < bc.k; ++j) {   // distinct slots, ending reads local
					ok = ok && bc.slot[j] < run.L + run.g && !((used >> bc.slot[j]) & 1u);
					used |= 1u << bc.slot[j];
					if (!((p.fwd_mask[c] >> j) & 1u)) ok = ok && bc.slot[j] < run.L;
				}
				for (uint32_t q = 0; q < row_n_end; ++q) {
					const uint32_t es = sp.ped ? (uint32_t)(q < 3u ? bc.slot[25 + q] : bc.pad[1]) : (sp.rows[run.row_off + i].end[q].info & 255u);
					ok = ok && sp.end_slots[sp.end_off[step.index] + ends + q] == es;
				}
				ends += row_n_end;
			}
			ok = ok && ends == run.n_ends;
			s.max_run_columns = std::max<uint64_t>(s.max_run_columns, run.ncols);
			s.max_workgroups = std::max<uint64_t>(s.max_workgroups, 1ull << (run.g - run.half));
			if (sp.ped) {
				const PedSlotExtra& ex = sp.pextra[step.index];
				s.max_lds_bytes = std::max<uint64_t>(s.max_lds_bytes, pedslot_lds_bytes(run.threads, run.ncols, ex));
				if (pslot_is_fact(ex.nf)) s.n_fact_runs++;
				s.backtrace_bytes += ((uint64_t)ex.rec_words * 4) << run.g;
			} else {
				s.max_lds_bytes = std::max<uint64_t>(s.max_lds_bytes, slot_run_lds_bytes(run.threads, run.lr, run.ncols));
				s.backtrace_bytes += (uint64_t)run.n_ends * run.threads * (1ull << (run.g - run.half));
			}
			if (run.half) s.n_halved_runs++;
			if (run.yflags & 1u) s.n_yform_runs++;
			s.n_resident_columns += run.ncols;
			s.n_vectorised_columns += run.ncols;
		}
		ok = ok && expect == p.n_cols && kc == sp.component_first_step.size();
		s.invariants_ok = ok ? 1 : 0;
		*out = s;
		return WHAMD_OK;
	}
	ResidentPlan plan;
	plan_forward(p, mode == "auto" || mode == "resident", 11, true, plan);
	whamd_plan_summary s{};
	s.n_columns = p.n_cols;
	s.n_steps = plan.steps.size();
	s.n_runs = plan.segments.size();
	s.max_coverage = p.max_k;
	s.n_components = plan.component_first_step.size();
	bool ok = true;
	// components: boundaries are exactly the steps whose first column no read enters, and no run spans one
	if (!plan.component_first_step.empty()) {
		size_t k = 0;
		for (size_t si = 0; si < plan.steps.size(); ++si) {
			const Step& step = plan.steps[si];
			const uint32_t c0 = step.kind == 1 ? plan.segments[step.index].c0 : step.index;
			const bool boundary = si == 0 || p.b[c0] == 0;
			if (boundary) { ok = ok && k < plan.component_first_step.size() && plan.component_first_step[k] == si; ++k; }
			if (step.kind == 1)
				for (uint32_t i = 1; i < plan.segments[step.index].ncols; ++i) ok = ok && p.b[c0 + i] != 0;
		}
		ok = ok && k == plan.component_first_step.size();
	}
	std::vector<uint8_t> seen(p.n_cols, 0);
	uint32_t expect = 0;
	for (const Step& step : plan.steps) {
		if (step.kind == 0) {
			ok = ok && step.index == expect && step.index < p.n_cols;
			if (step.index < p.n_cols) seen[step.index]++;
			expect = step.index + 1;
			continue;
		}
		const ResSegment& sg = plan.segments[step.index];
		ok = ok && sg.c0 == expect && sg.ncols >= 2 && sg.ncols <= (uint32_t)RES_MAXCOLS && sg.g <= (uint32_t)RES_GMAX;
		expect = sg.c0 + sg.ncols;
		uint64_t stage = 0;
		for (uint32_t i = 0; i < sg.ncols; ++i) {
			const uint32_t c = sg.c0 + i;
			if (c >= p.n_cols) { ok = false; break; }
			seen[c]++;
			const ResColumn& rc = plan.columns[sg.col_off + i];
			ok = ok && plan.col_to_res[c] == (int32_t)(sg.col_off + i);
			ok = ok && rc.Lb == p.b[c] - sg.g && rc.Lf == p.f[c] - sg.g && rc.ebits == (uint32_t)p.k[c] - p.f[c];
			const uint32_t lmax = sg.kind == 1 ? (uint32_t)PED_LMAX : (uint32_t)RES_LMAX;
			ok = ok && rc.Lb <= lmax && rc.Lf <= lmax && rc.ebits <= (uint32_t)RES_EMAX;
			ok = ok && rc.stage_off == stage;
			stage += sg.kind == 1 ? (uint64_t)rc.nwords : (uint64_t)rc.ebits * rc.nwords;
			if (rc.mode == RES_MODE_FOLDED) { s.n_folded_columns++; ok = ok && i + 1 < sg.ncols && rc.ebits == 0; }
			if (rc.mode != RES_MODE_GENERIC) s.n_vectorised_columns++;
			if (rc.nfold) ok = ok && rc.nfold <= RES_MAXFOLD && i >= rc.nfold;
			ok = ok && c + 1 < p.n_cols;  // the last column never runs resident
		}
		ok = ok && stage == sg.stage_words;
		const uint64_t lds = sg.kind == 1
			? (((uint64_t)sg.ncols * (PED_LDSWORDS + PED_TABLE) + (uint64_t)sg.n_terms * 2 + 3) & ~3ull) * 4 + 2 * (16ull << sg.max_l) + (uint64_t)sg.stage_words * 8
			: (uint64_t)sg.ncols * (64 + RES_TABLE) * 4 + 2 * (4ull << sg.max_l) + (uint64_t)sg.stage_words * 8;
		ok = ok && lds <= 160 * 1024;
		s.max_lds_bytes = std::max<uint64_t>(s.max_lds_bytes, lds);
		s.max_run_columns = std::max<uint64_t>(s.max_run_columns, sg.ncols);
		s.max_workgroups = std::max<uint64_t>(s.max_workgroups, 1ull << (sg.g - sg.half));  // launched workgroups
		if (sg.half) s.n_halved_runs++;
		s.n_resident_columns += sg.ncols;
		s.backtrace_bytes += (uint64_t)sg.stage_words * (1ull << sg.g) * 8;
	}
	ok = ok && expect == p.n_cols;
	for (uint32_t c = 0; c < p.n_cols; ++c) ok = ok && seen[c] == 1;
	s.invariants_ok = ok ? 1 : 0;
	*out = s;
	return WHAMD_OK;
	});
}

#ifdef WHAMD_DEBUG_BUILD   // libwhatshap_amd_debug.so only (include/whatshap_amd_debug.h): test infrastructure
whamd_status_t whamd_debug_emulate_slot_plan(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                            const whamd_pedigree_view* pedigree, int distrust_genotypes, const uint32_t* positions,
                                            size_t n_positions, int slot_l, int symmetry, uint32_t* index_out, uint32_t* score_out,
                                            uint64_t* n_run_columns_out) {
	return guarded([&]() -> whamd_status_t {
	const int lr = slot_l >= 200 ? 1 : slot_l >= 100 ? 3 : 2;   // slot_l + 100: 8 cells per thread, + 200: 2 cells per thread
	if (slot_l >= 100) slot_l -= slot_l >= 200 ? 200 : 100;
	Problem p;
	std::string msg;
	whamd_status_t st = build_problem(readset, recombcost, n_recombcost, pedigree, distrust_genotypes != 0, positions,
	                                  n_positions, p, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	SlotPlan sp;
	if (p.T != 1 || !plan_forward_slots(p, slot_l, symmetry, sp, lr)) return fail(WHAMD_ERR_UNSUPPORTED, "slot runs apply to a single individual only");
	std::vector<uint32_t> path;
	uint32_t score = 0;
	if (!emulate_slot_plan(p, sp, path, score, msg)) return fail(WHAMD_ERR_INVALID, "slot plan inconsistent: " + msg);
	if (index_out && !path.empty()) std::memcpy(index_out, path.data(), path.size() * sizeof(uint32_t));
	if (score_out) *score_out = score;
	if (n_run_columns_out) *n_run_columns_out = sp.n_run_columns;
	return WHAMD_OK;
	});
}

whamd_status_t whamd_debug_emulate_pedslot_plan(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                               const whamd_pedigree_view* pedigree, int distrust_genotypes, const uint32_t* positions,
                                               size_t n_positions, int slot_l, uint32_t* index_out, uint32_t* transmission_out,
                                               uint32_t* score_out, uint64_t* n_run_columns_out) {
	return guarded([&]() -> whamd_status_t {
	Problem p;
	std::string msg;
	whamd_status_t st = build_problem(readset, recombcost, n_recombcost, pedigree, distrust_genotypes != 0, positions,
	                                  n_positions, p, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	SlotPlan sp;
	if (p.T == 1 || !plan_forward_slots(p, slot_l > 0 ? -slot_l : 0, 0, sp) || !sp.ped)
		return fail(WHAMD_ERR_UNSUPPORTED, "pedigree slot runs apply to tables with one or two trios whose columns need at most 4 cost forms per transmission value");
	std::vector<uint32_t> path, trans;
	uint32_t score = 0;
	if (!emulate_pedslot_plan(p, sp, path, trans, score, msg)) return fail(WHAMD_ERR_INVALID, "pedigree slot plan inconsistent: " + msg);
	if (index_out && !path.empty()) std::memcpy(index_out, path.data(), path.size() * sizeof(uint32_t));
	if (transmission_out && !trans.empty()) std::memcpy(transmission_out, trans.data(), trans.size() * sizeof(uint32_t));
	if (score_out) *score_out = score;
	if (n_run_columns_out) *n_run_columns_out = sp.n_run_columns;
	return WHAMD_OK;
	});
}
whamd_status_t whamd_debug_lazy_terms_check(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                            const whamd_pedigree_view* pedigree, int distrust_genotypes, const uint32_t* positions, size_t n_positions,
                                            const uint8_t* need_in, int rounds, int* lazy_out, uint64_t* differences_out, uint64_t* built_before_out,
                                            uint64_t* built_after_out) {
	return guarded([&]() -> whamd_status_t {
	Problem eager, lazy;
	std::string msg;
	whamd_status_t st = build_problem(readset, recombcost, n_recombcost, pedigree, distrust_genotypes != 0, positions, n_positions, eager, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	st = build_problem(readset, recombcost, n_recombcost, pedigree, distrust_genotypes != 0, positions, n_positions, lazy, msg, false, /*lazy_fact_terms=*/true);
	if (st != WHAMD_OK) return fail(st, msg);
	if (lazy_out) *lazy_out = lazy.lazy_terms ? 1 : 0;
	const uint32_t n = eager.n_cols;
	uint64_t before = 0, after = 0, diff = 0;
	for (uint32_t c = 0; c < n; ++c) before += lazy.term_end(c, lazy.T - 1) > lazy.term_begin(c, 0);
	std::vector<uint8_t> need(n, 1);
	if (need_in) for (uint32_t c = 0; c < n; ++c) need[c] = need_in[c] != 0;
	rounds = std::max(1, rounds);
	for (int r = 0; r < rounds; ++r) {
		std::vector<uint8_t> part(n, 0);
		for (uint32_t c = 0; c < n; ++c) part[c] = need[c] && (int)(c % (uint32_t)rounds) == r;
		st = fill_lazy_terms(lazy, part, msg);
		if (st != WHAMD_OK) return fail(st, msg);
	}
	for (uint32_t c = 0; c < n; ++c) {
		after += lazy.term_end(c, lazy.T - 1) > lazy.term_begin(c, 0);
		if (!need[c]) continue;
		bool same = true;
		for (uint32_t t = 0; t < eager.T && same; ++t) {
			const uint64_t a0 = eager.term_begin(c, t), a1 = eager.term_end(c, t), b0 = lazy.term_begin(c, t), b1 = lazy.term_end(c, t);
			same = a1 - a0 == b1 - b0;
			for (uint64_t i = 0; same && i < a1 - a0; ++i)
				same = eager.terms[a0 + i].c == lazy.terms[b0 + i].c && eager.terms[a0 + i].plus == lazy.terms[b0 + i].plus && eager.terms[a0 + i].minus == lazy.terms[b0 + i].minus;
		}
		diff += !same;
	}
	// what does not depend on the route: the factorised line itself and the problem's shape
	if (eager.fterm_kind != lazy.fterm_kind || eager.fterms.size() != lazy.fterms.size()) diff += n;
	else for (size_t i = 0; i < eager.fterms.size(); ++i) if (eager.fterms[i].c != lazy.fterms[i].c || eager.fterms[i].plus != lazy.fterms[i].plus || eager.fterms[i].minus != lazy.fterms[i].minus) { ++diff; break; }
	if (lazy.value_bound < eager.value_bound) diff += n;   // (the lazy bound may only be LARGER: it is what rules 32-bit wrap-around out)
	if (differences_out) *differences_out = diff;
	if (built_before_out) *built_before_out = before;
	if (built_after_out) *built_after_out = after;
	return WHAMD_OK;
	});
}
#endif   // WHAMD_DEBUG_BUILD

}  // extern "C"

// A batch in flight is shared by its handles: the first whamd_pedmec_heuristic_wait on any of them collects all of them.
struct whamd_heuristic;
struct HeuristicBatchState {
	whamd::HeurBatch batch;
	std::vector<whamd_heuristic*> members;
	bool collected = false;
	whamd_status_t status = WHAMD_OK;
	std::string message;
	double t_enqueued = 0.0;
};

struct whamd_heuristic {
	whamd::HeurPlan plan;
	whamd::HeurResult result;
	whamd_heuristic_stats stats{};
	std::shared_ptr<HeuristicBatchState> batch;   // non-null while in flight
	bool finished = false;
};

namespace {
whamd_status_t heuristic_collect(whamd_heuristic* h) {
	if (h->finished) return WHAMD_OK;
	if (!h->batch) return fail(WHAMD_ERR_INVALID, "whamd_pedmec_heuristic_enqueue has not run");
	std::shared_ptr<HeuristicBatchState> st = h->batch;
	if (!st->collected) {
		st->collected = true;
		std::vector<whamd::HeurResult> results(st->members.size());
		st->status = st->batch.wait(results.data(), st->message);
		if (st->status == WHAMD_OK) {
			const double t_done = now_ms();
			// allele votes + phasing per column (host, src/pedmecheuristic.cpp:361-406) of all tables at once
			whamd::parallel_ranges(st->members.size(), whamd::host_threads(st->members.size(), 1), [&](uint64_t i0, uint64_t i1, uint32_t) {
				for (uint64_t i = i0; i < i1; ++i) {
					whamd_heuristic* m = st->members[i];
					const double t0 = now_ms();
					m->result = std::move(results[i]);
					whamd::heuristic_finish(m->plan, m->result);
					m->stats.max_solutions = m->result.max_solutions;
					m->stats.total_solutions = m->result.total_solutions;
					m->stats.device_ms = m->result.device_ms;
					m->stats.host_prepare_ms += std::max(0.0, (t_done - st->t_enqueued) - m->result.device_ms);
					m->stats.host_finish_ms = now_ms() - t0;
				}
			});
		}
		for (whamd_heuristic* m : st->members) { m->finished = st->status == WHAMD_OK; m->batch.reset(); }
	}
	if (st->status != WHAMD_OK) return fail(st->status, st->message);
	return WHAMD_OK;
}

whamd_status_t heuristic_enqueue_jobs(const whamd_heuristic_job* jobs, size_t n_jobs, int device, whamd_heuristic** out) {
	if (!out || (!jobs && n_jobs)) return fail(WHAMD_ERR_INVALID, "null argument");
	for (size_t i = 0; i < n_jobs; ++i) out[i] = nullptr;
	std::vector<std::unique_ptr<whamd_heuristic>> hs(n_jobs);
	std::vector<whamd_status_t> status(n_jobs, WHAMD_OK);
	std::vector<std::string> messages(n_jobs);
	std::vector<double> prepare_ms(n_jobs, 0.0);
	// the plans (flattening + the per-column bookkeeping of solve() that does not depend on the beam): a few host threads
	whamd::parallel_ranges(n_jobs, whamd::host_threads(n_jobs, 1), [&](uint64_t i0, uint64_t i1, uint32_t) {
		for (uint64_t i = i0; i < i1; ++i) {
			const double t0 = now_ms();
			hs[i].reset(new whamd_heuristic());
			const whamd_heuristic_job& j = jobs[i];
			status[i] = whamd::build_heuristic_plan(j.readset, j.recombcost, j.n_recombcost, j.pedigree, j.distrust_genotypes != 0, j.positions, j.n_positions,
			                                        j.row_limit, j.allow_mutations != 0, hs[i]->plan, messages[i]);
			prepare_ms[i] = now_ms() - t0;
		}
	});
	for (size_t i = 0; i < n_jobs; ++i) if (status[i] != WHAMD_OK) return fail(status[i], messages[i]);
	std::shared_ptr<HeuristicBatchState> st(new HeuristicBatchState());
	std::vector<const whamd::HeurPlan*> plans(n_jobs);
	for (size_t i = 0; i < n_jobs; ++i) plans[i] = &hs[i]->plan;
	std::string msg;
	st->t_enqueued = now_ms();
	const whamd_status_t es = st->batch.enqueue(plans.data(), n_jobs, device, msg);
	if (es != WHAMD_OK) return fail(es, msg);
	for (size_t i = 0; i < n_jobs; ++i) {
		whamd_heuristic* h = hs[i].get();
		h->stats.n_columns = h->plan.n_cols; h->stats.n_reads = h->plan.n_reads; h->stats.n_samples = h->plan.n_samples; h->stats.row_limit = h->plan.row_limit;
		h->stats.host_prepare_ms = prepare_ms[i];
		h->batch = st;
		st->members.push_back(h);
	}
	for (size_t i = 0; i < n_jobs; ++i) out[i] = hs[i].release();
	return WHAMD_OK;
}

whamd_status_t heuristic_create_common(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost, const whamd_pedigree_view* pedigree,
                                       int distrust_genotypes, const uint32_t* positions, size_t n_positions, uint32_t row_limit, int allow_mutations,
                                       int device, bool on_host, whamd_heuristic** out) {
	if (!out) return fail(WHAMD_ERR_INVALID, "out is NULL");
	*out = nullptr;
	if (!on_host) {
		whamd_heuristic_job job{readset, recombcost, n_recombcost, pedigree, distrust_genotypes, positions, n_positions, row_limit, allow_mutations};
		whamd_heuristic* h = nullptr;
		whamd_status_t st = heuristic_enqueue_jobs(&job, 1, device, &h);
		if (st != WHAMD_OK) return st;
		st = heuristic_collect(h);
		if (st != WHAMD_OK) { delete h; return st; }
		*out = h;
		return WHAMD_OK;
	}
#ifndef WHAMD_DEBUG_BUILD
	return fail(WHAMD_ERR_DEVICE, "the host instantiation of the heuristic exists in the debug library only");
#else
	const double t0 = now_ms();
	std::unique_ptr<whamd_heuristic> h(new whamd_heuristic());
	std::string msg;
	whamd_status_t st = whamd::build_heuristic_plan(readset, recombcost, n_recombcost, pedigree, distrust_genotypes != 0, positions, n_positions, row_limit,
	                                                allow_mutations != 0, h->plan, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	const double t1 = now_ms();
	st = whamd::heuristic_solve_host(h->plan, h->result, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	const double t2 = now_ms();
	whamd::heuristic_finish(h->plan, h->result);
	h->stats.n_columns = h->plan.n_cols; h->stats.n_reads = h->plan.n_reads; h->stats.max_solutions = h->result.max_solutions;
	h->stats.total_solutions = h->result.total_solutions; h->stats.device_ms = h->result.device_ms;
	h->stats.host_prepare_ms = (t1 - t0) + ((t2 - t1) - h->result.device_ms); h->stats.host_finish_ms = now_ms() - t2;
	h->stats.n_samples = h->plan.n_samples; h->stats.row_limit = h->plan.row_limit;
	h->finished = true;
	*out = h.release();
	return WHAMD_OK;
#endif
}
}  // namespace

extern "C" {

whamd_status_t whamd_pedmec_heuristic_create(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                             const whamd_pedigree_view* pedigree, int distrust_genotypes, const uint32_t* positions, size_t n_positions,
                                             uint32_t row_limit, int allow_mutations, int device, whamd_heuristic** out) {
	return guarded([&]() -> whamd_status_t {
	return heuristic_create_common(readset, recombcost, n_recombcost, pedigree, distrust_genotypes, positions, n_positions, row_limit, allow_mutations, device, false, out);
	});
}

whamd_status_t whamd_pedmec_heuristic_enqueue_many(const whamd_heuristic_job* jobs, size_t n_jobs, int device, whamd_heuristic** out) {
	return guarded([&]() -> whamd_status_t {
	return heuristic_enqueue_jobs(jobs, n_jobs, device, out);
	});
}

whamd_status_t whamd_pedmec_heuristic_wait(whamd_heuristic* h) {
	return guarded([&]() -> whamd_status_t {
	if (!h) return fail(WHAMD_ERR_INVALID, "null argument");
	return heuristic_collect(h);
	});
}

#ifdef WHAMD_DEBUG_BUILD
whamd_status_t whamd_debug_pedmec_heuristic_create_host(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                                        const whamd_pedigree_view* pedigree, int distrust_genotypes, const uint32_t* positions,
                                                        size_t n_positions, uint32_t row_limit, int allow_mutations, whamd_heuristic** out) {
	return guarded([&]() -> whamd_status_t {
	return heuristic_create_common(readset, recombcost, n_recombcost, pedigree, distrust_genotypes, positions, n_positions, row_limit, allow_mutations, 0, true, out);
	});
}
#endif

uint64_t whamd_pedmec_heuristic_column_count(const whamd_heuristic* h) { return h ? h->plan.n_cols : 0; }
uint32_t whamd_pedmec_heuristic_sample_count(const whamd_heuristic* h) { return h ? h->plan.n_samples : 0; }
uint32_t whamd_pedmec_heuristic_read_count(const whamd_heuristic* h) { return h ? h->plan.n_reads : 0; }

whamd_status_t whamd_pedmec_heuristic_get(const whamd_heuristic* h, float* score, uint8_t* bipartition, uint32_t* transmission, int8_t* haplotypes,
                                          uint8_t* mutated, uint32_t* sample_ids, uint32_t* positions) {
	if (!h) return fail(WHAMD_ERR_INVALID, "null argument");
	if (!h->finished) return fail(WHAMD_ERR_INVALID, "the solve is still in flight: call whamd_pedmec_heuristic_wait first");
	if (score) *score = h->result.score;
	if (bipartition && !h->result.bipartition.empty()) std::memcpy(bipartition, h->result.bipartition.data(), h->result.bipartition.size());
	if (transmission && !h->result.transmission.empty()) std::memcpy(transmission, h->result.transmission.data(), h->result.transmission.size() * 4);
	if (haplotypes && !h->result.haplotypes.empty()) std::memcpy(haplotypes, h->result.haplotypes.data(), h->result.haplotypes.size());
	if (mutated && !h->result.mutated.empty()) std::memcpy(mutated, h->result.mutated.data(), h->result.mutated.size());
	if (sample_ids && !h->plan.sample_global_id.empty()) std::memcpy(sample_ids, h->plan.sample_global_id.data(), h->plan.sample_global_id.size() * 4);
	if (positions && !h->plan.positions.empty()) std::memcpy(positions, h->plan.positions.data(), h->plan.positions.size() * 4);
	return WHAMD_OK;
}

whamd_status_t whamd_pedmec_heuristic_get_stats(const whamd_heuristic* h, whamd_heuristic_stats* stats_out) {
	if (!h || !stats_out) return fail(WHAMD_ERR_INVALID, "null argument");
	*stats_out = h->stats;
	return WHAMD_OK;
}

void whamd_pedmec_heuristic_destroy(whamd_heuristic* h) {
	if (h && h->batch) {   // still in flight: the batch reads this handle's plan -- collect it first (the other members keep their results)
		try {
			(void)heuristic_collect(h);   // allocates, runs host threads, waits for the device: nothing it throws may cross the C boundary
		} catch (...) {
		}
	}
	delete h;
}

// std::hash tie-break of ReadSet::sort (src/readset.h:52-55,78-82): exported so that the Python mirror of
// ReadSet.sort() orders reads exactly like the reference built against the same libstdc++.
uint64_t whamd_read_sort_hash(const char* name, int source_id) {
	return (uint64_t)(std::hash<std::string>()(std::string(name)) ^ std::hash<int>()(source_id));
}

whamd_status_t whamd_genotype_likelihoods(const whamd_readset_view* readset, const uint32_t* recombcost, size_t n_recombcost,
                                          const whamd_pedigree_view* pedigree, const uint32_t* positions, size_t n_positions,
                                          int device, uint32_t window, double* gl_out, size_t gl_capacity,
                                          whamd_genotype_stats* stats_out) {
	return guarded([&]() -> whamd_status_t {
	const double t0 = now_ms();
	Problem p;
	std::string msg;
	whamd_status_t st = build_problem(readset, recombcost, n_recombcost, pedigree, false, positions, n_positions, p, msg, /*columns_only=*/true);
	if (st != WHAMD_OK) return fail(st, msg);
	const size_t need = (size_t)p.n_ind * p.n_cols * 3;
	if (need && (!gl_out || gl_capacity < need)) return fail(WHAMD_ERR_INVALID, "gl_out holds fewer than individuals * columns * 3 values");
	GenotypeModel model;
	st = build_genotype_model(p, model, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	std::vector<double> gl;
	GenotypeStats gs;
	st = genotype_solve_device(p, model, device, window, gl, gs, msg);
	if (st != WHAMD_OK) return fail(st, msg);
	if (need) std::memcpy(gl_out, gl.data(), need * sizeof(double));
	if (stats_out) {
		whamd_genotype_stats o{};
		o.n_columns = gs.n_columns; o.n_cells = gs.n_cells; o.launches = gs.launches;
		o.backward_ms = gs.backward_ms; o.forward_ms = gs.forward_ms; o.total_ms = gs.total_ms;
		o.host_prepare_ms = (now_ms() - t0) - gs.total_ms;
		o.window = gs.window; o.max_coverage = gs.max_coverage; o.transmissions = gs.transmissions; o.slot_runs = gs.slot_runs;
		*stats_out = o;
	}
	return WHAMD_OK;
	});
}

uint64_t whamd_host_pool_idle_bytes(void) { return (uint64_t)whamd::host_pool_idle_bytes(); }

void whamd_release_caches(void) {
	genotype_release_cache();
	whamd::heuristic_release_cache();
	whamd::dptable_release_caches();
	whamd::host_pool_release();   // the host side's kept blocks (host_memory.cpp)
}

}  // extern "C"
