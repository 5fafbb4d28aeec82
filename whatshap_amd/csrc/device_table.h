// device_table.h -- host-visible interface of the device driver (dp_device.hip).
#pragma once
#include <string>

#include "device_types.h"
#include "problem.h"

namespace whamd {

class DeviceTable {
public:
	DeviceTable();
	~DeviceTable();
	DeviceTable(const DeviceTable&) = delete;
	DeviceTable& operator=(const DeviceTable&) = delete;

	static int device_count();
	static bool device_pci_bus_id(int device, std::string& out);   // "0000:c5:00.0"; false if there is no such device
	// Builds the per-column descriptors for `p` and uploads everything the kernels read.
	// (`p` is not const: a table with lazy generic terms -- Problem::lazy_terms -- gets the term lists of the columns its plan leaves outside runs here)
	whamd_status_t upload(Problem& p, int device, std::string& msg);
	// Forward pass + backtrace on the device; fills s.path_*, s.optimal_score and the timing fields of st.
	whamd_status_t solve(const Problem& p, Solution& s, whamd_solve_stats& st, std::string& msg);
	// The two halves of solve(): enqueue() only submits the launches to the table's stream (several tables can be in
	// flight at once on one device), wait() blocks until the path has arrived and reads the event timings.
	whamd_status_t enqueue(const Problem& p, Solution& s, std::string& msg);
	// Resumable enqueue(): at most `budget` forward launches per call; `done` once backtrace and downloads are submitted.
	whamd_status_t enqueue_some(const Problem& p, Solution& s, uint64_t budget, bool& done, std::string& msg);
	whamd_status_t wait(const Problem& p, Solution& s, whamd_solve_stats& st, std::string& msg);
	// forward_ms / backtrace_ms / total_ms of the solve wait() collected last: read from the events on demand (wait() leaves them 0).
	void read_timing(whamd_solve_stats& st);
	// Several tables of ONE device as one sequence of launches (see dp_device.hip, "group solve"): every table must be group_eligible().
	// Afterwards each table is in flight exactly as after enqueue(): collect with wait().
	static whamd_status_t enqueue_group(DeviceTable* const* tables, const Problem* const* problems, Solution* const* solutions, size_t n_tables, std::string& msg);
	// Before the wait() of several tables in flight: one host wait per stream that carries tails of a group (dp_device.hip).
	static void wait_last_of_each_stream(DeviceTable* const* tables, size_t n_tables);
	bool group_eligible(const Problem& p) const;
	int device_index() const;
	uint32_t widest_launch() const;   // workgroups of the widest launch of the schedule
	// Drops a partially submitted solve (enqueue_some that has not reported `done`): drains the stream, rewinds the cursor.
	void abort_enqueue();
	// Frees the device buffers, the stream and the events; the next upload() recreates them.
	void release_device();
	// Solver variant ("auto", "column", "column_keys", "resident", "slots"); takes effect at the next upload().
	// auto: slot runs (slots.h) for a single individual, LDS-resident runs for a trio, per-column kernels otherwise.
	bool set_path(const std::string& path);
	// Preferred log2 slice size of the resident path (tuning knob); takes effect at the next upload().
	void set_l_pref(int l);
	// Preferred number of local slots of a slot run (9 .. 12: 1 .. 8 waves per workgroup); next upload().
	void set_slot_l(int l);
	// Reg slots of a slot run: 2 (4 cells per thread, default) or 3 (8 cells per thread); next upload().
	void set_slot_lr(int lr);
	// Fold columns in which no read ends into the next resident column (default on); next upload().
	void set_fold(bool v);
	// Exploit D[~x] == D[x] in single-individual runs: 0 off, 1 full-chip runs (default), 2 every run; next upload().
	void set_symmetry(int level);
	// Lanes over which the connected components of a single-individual table are spread (default 32, 1 = off); the
	// lanes advance in lockstep, their runs go out as batched launches; next upload().
	void set_lanes(int n);
	// Upper bound of the backtrace arena in bytes (0 = whatever free HBM allows).  A table whose records need more is
	// solved in windows (the forward pass of every window but the newest runs twice); next upload().
	void set_arena_limit(uint64_t bytes);
	// The table will be solved together with many others (whamd_dptable_enqueue_many): a wide single-individual table then plans eight cells
	// per thread and twelve local slots (half the wavefronts per table); next upload().
	void set_shared_launches(bool v);
	// the table's launches will run BESIDE other tables' on the same device (own streams, whamd_dptable_enqueue_many): kernels that leave room on a CU
	void set_side_by_side(bool v);

private:
	whamd_status_t enqueue_some_unguarded(const Problem& p, Solution& s, uint64_t budget, bool& done, std::string& msg);
	struct Impl;
	Impl* impl_;
};

// process-wide caches of the device driver (the arena of the table closed last, the pinned upload staging area): whamd_release_caches
void dptable_release_caches();

}  // namespace whamd
