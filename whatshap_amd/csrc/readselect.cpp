// readselect.cpp -- read selection: which reads of a ReadSet are kept so that no variant is
// covered more than max_cov times, most informative reads first.  Host code: the work is a priority queue and a few
// hash sets, sequential by construction (every selection changes the scores the next one is taken by).
//
// Behaviour follows whatshap/readselect.pyx:
//   readselection            :218-255   two passes (preferred sources first), both through readselection_helper
//   readselection_helper     :160-215   slices until no read is undecided; bridging reads between the blocks of a slice
//   _slice_read_selection    :103-157   pop best read; keep it if it covers a new variant and coverage allows it; lower
//                                       the score of every read that shares a newly covered variant
//   _compute_score_for_read  :56-88     (good - bad, good - bad, min quality)
//   _update_score_for_reads  :38-53     first component minus the read's variants that are NOT among the newly covered ones
//   PriorityQueue            whatshap/priorityqueue.pyx:52-190 (binary max-heap, lexicographic score, recursive sifts)
//   CovMonitor               whatshap/coverage.py:1-14
//   ComponentFinder          whatshap/graph.py:10-83 (union-find, smallest value represents the component)
//
// WHICH read wins a tie depends on the order in which equal scores entered and moved through the heap, and in the
// reference that order comes from iterating Python sets of read indices (readselect.pyx:97 `for index in read_indices`,
// :150 `for element in d_set`) and one std::unordered_set<int> (:142).  CPython's set is an open-addressing table
// (Objects/setobject.c of CPython 3.10: hash(int) == int, linear probes of 9 then i*5+1+perturb, growth x4 up to 50000
// entries and x2 beyond, copies and differences built in table order), so the order is a deterministic function of the
// operations performed; PySetInt below replays exactly those operations, and the unordered_set is the same libstdc++ type
// fed the same sequence.  tests/test_readselect.py compares the selection with the built reference module on tie-heavy
// inputs.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/whatshap_amd.h"

namespace whamd {
void set_last_error(const std::string& msg);
}

namespace {

// ---- CPython 3.10 set restricted to non-negative ints (hash(i) == i) -----------------------------------------------------
class PySetInt {
public:
	static constexpr int64_t EMPTY = -1, DUMMY = -2;
	static constexpr size_t MINSIZE = 8, LINEAR_PROBES = 9, PERTURB_SHIFT = 5;

	PySetInt() : tab_(MINSIZE, EMPTY), mask_(MINSIZE - 1) {}

	size_t size() const { return used_; }

	// set_add_entry
	void add(int64_t key) {
		size_t perturb = (size_t)key, i = (size_t)key & mask_;
		int64_t* freeslot = nullptr;
		int64_t* entry;
		for (;;) {
			entry = &tab_[i];
			size_t probes = (i + LINEAR_PROBES <= mask_) ? LINEAR_PROBES : 0;
			do {
				if (*entry == EMPTY) goto found_unused_or_dummy;
				if (*entry == key) return;
				if (*entry == DUMMY) freeslot = entry;
				++entry;
			} while (probes--);
			perturb >>= PERTURB_SHIFT;
			i = (i * 5 + 1 + perturb) & mask_;
		}
	found_unused_or_dummy:
		if (freeslot) { ++used_; *freeslot = key; return; }
		++fill_; ++used_;
		*entry = key;
		if (fill_ * 5 < mask_ * 3) return;
		resize(used_ > 50000 ? used_ * 2 : used_ * 4);
	}

	bool contains(int64_t key) const { return lookup(key) != nullptr; }

	// set_discard_entry: the entry becomes a dummy, the table is never shrunk
	bool discard(int64_t key) {
		int64_t* e = const_cast<int64_t*>(lookup(key));
		if (!e) return false;
		*e = DUMMY;
		--used_;
		return true;
	}

	// tail of set_difference_update_internal (`s -= other`, and the copy-and-discard form of difference()): "if more than
	// 1/4th are dummies, then resize them away" -- the table may SHRINK below the largest key, after which the iteration
	// order is no longer ascending.  set.remove() / set.discard() alone never do this.
	void finish_difference_update() {
		if (fill_ - used_ <= mask_ / 4) return;
		resize(used_ > 50000 ? used_ * 2 : used_ * 4);
	}

	// set.update(list)
	void update(const int32_t* keys, size_t n) { for (size_t j = 0; j < n; ++j) add(keys[j]); }

	// set(range(n))
	static PySetInt from_range(int64_t n) { PySetInt s; for (int64_t i = 0; i < n; ++i) s.add(i); return s; }

	// set(other) / other.copy(): make_new_set -> set_update_internal -> set_merge into an empty set
	PySetInt copy() const {
		PySetInt r;
		if (used_ == 0) return r;
		if ((r.fill_ + used_) * 5 >= r.mask_ * 3) r.resize((r.used_ + used_) * 2);
		if (r.mask_ == mask_ && fill_ == used_) {
			r.tab_ = tab_;
			r.fill_ = fill_;
			r.used_ = used_;
			return r;
		}
		r.fill_ = used_;
		r.used_ = used_;
		for (int64_t key : tab_) if (key >= 0) insert_clean(r.tab_, r.mask_, key);
		return r;
	}

	// self.difference(other) where `other` is a set: other_size = len(other), in_other = membership test
	template <class InOther, class OtherItems>
	PySetInt difference(size_t other_size, InOther in_other, OtherItems for_each_other) const {
		if ((used_ >> 2) > other_size) {          // set_copy_and_difference
			PySetInt r = copy();
			for_each_other([&](int64_t key) { r.discard(key); });
			r.finish_difference_update();
			return r;
		}
		PySetInt r;
		for (int64_t key : tab_) if (key >= 0 && !in_other(key)) r.add(key);
		return r;
	}

	template <class F> void for_each(F f) const { for (int64_t key : tab_) if (key >= 0) f(key); }

private:
	const int64_t* lookup(int64_t key) const {       // set_lookkey
		size_t perturb = (size_t)key, i = (size_t)key & mask_;
		for (;;) {
			const int64_t* entry = &tab_[i];
			size_t probes = (i + LINEAR_PROBES <= mask_) ? LINEAR_PROBES : 0;
			do {
				if (*entry == EMPTY) return nullptr;
				if (*entry == key) return entry;
				++entry;
			} while (probes--);
			perturb >>= PERTURB_SHIFT;
			i = (i * 5 + 1 + perturb) & mask_;
		}
	}

	static void insert_clean(std::vector<int64_t>& tab, size_t mask, int64_t key) {
		size_t perturb = (size_t)key, i = (size_t)key & mask;
		for (;;) {
			int64_t* entry = &tab[i];
			size_t probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
			do {
				if (*entry == EMPTY) { *entry = key; return; }
				++entry;
			} while (probes--);
			perturb >>= PERTURB_SHIFT;
			i = (i * 5 + 1 + perturb) & mask;
		}
	}

	void resize(size_t minused) {                   // set_table_resize
		size_t newsize = MINSIZE;
		while (newsize <= minused) newsize <<= 1;
		if (newsize == MINSIZE && tab_.size() == MINSIZE && fill_ == used_) return;   // small table without dummies: nothing to do
		std::vector<int64_t> fresh(newsize, EMPTY);
		for (int64_t key : tab_) if (key >= 0) insert_clean(fresh, newsize - 1, key);
		tab_.swap(fresh);
		mask_ = newsize - 1;
		fill_ = used_;
	}

	std::vector<int64_t> tab_;
	size_t mask_, fill_ = 0, used_ = 0;
};

// ---- whatshap/priorityqueue.pyx ------------------------------------------------------------------------------------------
struct Score {
	int32_t v[3];
};
inline bool score_lower(const Score& a, const Score& b) {
	for (int i = 0; i < 3; ++i) {
		if (a.v[i] < b.v[i]) return true;
		if (a.v[i] > b.v[i]) return false;
	}
	return false;
}

class Heap {
public:
	explicit Heap(size_t n_items) : pos_(n_items, -1) {}
	bool empty() const { return heap_.empty(); }
	void push(const Score& s, int32_t item) {
		heap_.push_back({s, item});
		pos_[item] = (int32_t)heap_.size() - 1;
		sift_up((int32_t)heap_.size() - 1);
	}
	std::pair<Score, int32_t> pop() {
		Entry first = heap_.front(), last = heap_.back();
		heap_.pop_back();
		pos_[first.item] = -1;
		if (!heap_.empty()) {
			heap_[0] = last;
			pos_[last.item] = 0;
			sift_down(0);
		}
		return {first.score, first.item};
	}
	const Score* score_of(int32_t item) const { return pos_[item] < 0 ? nullptr : &heap_[pos_[item]].score; }
	void change_score(int32_t item, const Score& s) {
		const int32_t p = pos_[item];
		const Score old = heap_[p].score;
		heap_[p].score = s;
		if (score_lower(old, s)) sift_up(p); else sift_down(p);
	}

private:
	struct Entry { Score score; int32_t item; };
	void swap_entries(int32_t a, int32_t b) {
		std::swap(heap_[a], heap_[b]);
		pos_[heap_[a].item] = a;
		pos_[heap_[b].item] = b;
	}
	void sift_up(int32_t i) {
		while (i > 0) {
			const int32_t parent = (i - 1) / 2;
			if (!score_lower(heap_[parent].score, heap_[i].score)) return;
			swap_entries(parent, i);
			i = parent;
		}
	}
	void sift_down(int32_t i) {
		const int32_t n = (int32_t)heap_.size();
		for (;;) {
			const int32_t l = 2 * i + 1, r = 2 * i + 2;
			int32_t child;
			if (r < n) child = score_lower(heap_[l].score, heap_[r].score) ? r : l;
			else if (l < n) child = l;
			else return;
			if (!score_lower(heap_[i].score, heap_[child].score)) return;
			swap_entries(child, i);
			i = child;
		}
	}
	std::vector<Entry> heap_;
	std::vector<int32_t> pos_;
};

// ---- whatshap/graph.py ComponentFinder over variant indices ---------------------------------------------------------------
// (the reference keys its nodes by position; positions and variant indices are in the same order, so "smallest value" is
//  the same node either way, and only the partition is ever looked at)
class Components {
public:
	explicit Components(size_t n) : parent_(n) { for (size_t i = 0; i < n; ++i) parent_[i] = (int32_t)i; }
	int32_t find(int32_t x) {
		int32_t root = x;
		while (parent_[root] != root) root = parent_[root];
		while (parent_[x] != root) { const int32_t next = parent_[x]; parent_[x] = root; x = next; }
		return root;
	}
	void merge(int32_t a, int32_t b) {
		a = find(a); b = find(b);
		if (a == b) return;
		if (a < b) parent_[b] = a; else parent_[a] = b;
	}
private:
	std::vector<int32_t> parent_;
};

struct Selection {
	const whamd_readset_view* rs;
	uint32_t n_reads, n_variants, max_cov;
	bool bridging;
	std::vector<int32_t> var_index;                 // [entries] variant index of every read entry (vcf_indices[position])
	std::vector<uint64_t> v2r_ptr;                  // variant_to_reads_map, CSR, reads in ascending index
	std::vector<int32_t> v2r;
	std::vector<Score> initial_score;               // _compute_score_for_read does not depend on the selection so far
	std::vector<uint32_t> coverage;                 // CovMonitor
	std::vector<uint8_t> selected;                  // selected_reads

	uint64_t first(uint32_t r) const { return rs->read_ptr[r]; }
	uint32_t count(uint32_t r) const { return (uint32_t)(rs->read_ptr[r + 1] - rs->read_ptr[r]); }

	uint32_t max_coverage(uint32_t begin, uint32_t end) const {
		uint32_t m = 0;
		for (uint32_t i = begin; i < end; ++i) m = std::max(m, coverage[i]);
		return m;
	}
	void add_read(uint32_t begin, uint32_t end) { for (uint32_t i = begin; i < end; ++i) ++coverage[i]; }

	Heap build_queue(const PySetInt& reads) const {   // _construct_priorityqueue
		Heap pq(n_reads);
		reads.for_each([&](int64_t r) { pq.push(initial_score[(size_t)r], (int32_t)r); });
		return pq;
	}

	// _slice_read_selection; in_slice / violating are [n_reads] flags, slice_order the reads of the slice as selected
	void slice(Heap& pq, std::vector<uint8_t>& in_slice, std::vector<int32_t>& slice_order, std::vector<int32_t>& violating) {
		std::vector<uint8_t> covered(n_variants, 0);   // already_covered_variants (by variant index: positions are distinct)
		std::unordered_set<int> fresh;                 // variants_covered_by_this_read: positions, iterated in libstdc++ order
		std::unordered_map<int, int32_t> index_of;     // position -> variant index, for the positions in `fresh`
		while (!pq.empty()) {
			fresh.clear();
			const int32_t item = pq.pop().second;
			const uint64_t e0 = first(item);
			const uint32_t cnt = count(item);
			bool covers_new = false;
			for (uint32_t i = 0; i < cnt; ++i) {
				if (covered[var_index[e0 + i]]) continue;
				covers_new = true;
				fresh.insert(rs->var_position[e0 + i]);
			}
			const uint32_t begin = (uint32_t)var_index[e0], end = (uint32_t)var_index[e0 + cnt - 1] + 1;
			if (max_coverage(begin, end) >= max_cov) { violating.push_back(item); continue; }
			if (!covers_new) continue;
			add_read(begin, end);
			in_slice[item] = 1;
			slice_order.push_back(item);
			index_of.clear();
			for (uint32_t i = 0; i < cnt; ++i) index_of[rs->var_position[e0 + i]] = var_index[e0 + i];
			PySetInt to_update;
			for (int pos : fresh) {
				const int32_t v = index_of[pos];
				covered[v] = 1;
				to_update.update(v2r.data() + v2r_ptr[v], (size_t)(v2r_ptr[v + 1] - v2r_ptr[v]));
			}
			const PySetInt d_set = to_update.difference(
				slice_order.size(), [&](int64_t r) { return in_slice[(size_t)r] != 0; },
				[&](auto discard) { for (int32_t r : slice_order) discard((int64_t)r); });
			d_set.for_each([&](int64_t r) {
				const Score* old = pq.score_of((int32_t)r);
				if (!old) return;
				Score s = *old;
				const uint64_t q0 = first((uint32_t)r);
				const uint32_t qn = count((uint32_t)r);
				for (uint32_t i = 0; i < qn; ++i) if (fresh.find(rs->var_position[q0 + i]) == fresh.end()) s.v[0] -= 1;
				pq.change_score((int32_t)r, s);
			});
		}
	}

	// readselection_helper
	void helper(PySetInt& undecided) {
		std::vector<uint8_t> in_slice(n_reads, 0);
		std::vector<int32_t> slice_order, violating;
		while (undecided.size() > 0) {
			Heap pq = build_queue(undecided);
			slice_order.clear();
			violating.clear();
			slice(pq, in_slice, slice_order, violating);
			for (int32_t r : slice_order) { selected[r] = 1; undecided.discard(r); }   // undecided_reads -= reads_in_slice
			undecided.finish_difference_update();
			for (int32_t r : violating) undecided.discard(r);                           // undecided_reads -= reads_violating_coverage
			undecided.finish_difference_update();
			Components comp(n_variants);
			for (int32_t r : slice_order) {
				const uint64_t e0 = first(r);
				for (uint32_t i = 1; i < count(r); ++i) comp.merge(var_index[e0], var_index[e0 + i]);
			}
			if (bridging) {
				Heap bq = build_queue(undecided);
				while (!bq.empty()) {
					const int32_t r = bq.pop().second;
					const uint64_t e0 = first(r);
					const uint32_t cnt = count(r);
					const int32_t block0 = comp.find(var_index[e0]);
					bool two_blocks = false;
					for (uint32_t i = 1; i < cnt; ++i) if (comp.find(var_index[e0 + i]) != block0) { two_blocks = true; break; }
					const uint32_t begin = (uint32_t)var_index[e0], end = (uint32_t)var_index[e0 + cnt - 1] + 1;
					if (max_coverage(begin, end) >= max_cov) { undecided.discard(r); continue; }
					if (!two_blocks) continue;
					selected[r] = 1;
					add_read(begin, end);
					undecided.discard(r);
					for (uint32_t i = 1; i < cnt; ++i) comp.merge(var_index[e0], var_index[e0 + i]);
				}
			}
			for (int32_t r : slice_order) in_slice[r] = 0;
		}
	}
};

}  // namespace

extern "C" whamd_status_t whamd_readselection(const whamd_readset_view* rs, const int32_t* read_source_id, const int32_t* preferred_source_ids,
                                              size_t n_preferred, uint32_t max_cov, int bridging, uint8_t* selected_out, uint64_t* n_selected) {
	if (!rs || !selected_out || (rs->n_reads && (!rs->read_ptr || !rs->var_position || !rs->var_quality)) || (n_preferred && (!preferred_source_ids || !read_source_id))) {
		whamd::set_last_error("whamd_readselection: null argument");
		return WHAMD_ERR_INVALID;
	}
	Selection s;
	s.rs = rs;
	s.n_reads = rs->n_reads;
	s.max_cov = max_cov;
	s.bridging = bridging != 0;
	const uint64_t entries = s.n_reads ? rs->read_ptr[s.n_reads] : 0;
	for (uint32_t r = 0; r < s.n_reads; ++r) {
		if (s.count(r) < 2) {          // readselect.pyx:236-239
			whamd::set_last_error("readselection expects reads that cover at least two variants");
			return WHAMD_ERR_INVALID;
		}
	}
	// ReadSet::get_positions (src/readset.cpp:54-62): sorted distinct positions
	std::vector<int32_t> positions(rs->var_position, rs->var_position + entries);
	std::sort(positions.begin(), positions.end());
	positions.erase(std::unique(positions.begin(), positions.end()), positions.end());
	s.n_variants = (uint32_t)positions.size();
	s.var_index.resize(entries);
	for (uint64_t e = 0; e < entries; ++e)
		s.var_index[e] = (int32_t)(std::lower_bound(positions.begin(), positions.end(), rs->var_position[e]) - positions.begin());
	s.v2r_ptr.assign((size_t)s.n_variants + 1, 0);
	for (uint64_t e = 0; e < entries; ++e) ++s.v2r_ptr[(size_t)s.var_index[e] + 1];
	for (uint32_t v = 0; v < s.n_variants; ++v) s.v2r_ptr[v + 1] += s.v2r_ptr[v];
	s.v2r.resize(entries);
	{
		std::vector<uint64_t> fill(s.v2r_ptr.begin(), s.v2r_ptr.end() - 1);
		for (uint32_t r = 0; r < s.n_reads; ++r)
			for (uint64_t e = rs->read_ptr[r]; e < rs->read_ptr[r + 1]; ++e) s.v2r[fill[s.var_index[e]]++] = (int32_t)r;
	}
	s.initial_score.resize(s.n_reads);
	for (uint32_t r = 0; r < s.n_reads; ++r) {
		const uint64_t e0 = s.first(r);
		const uint32_t cnt = s.count(r);
		int32_t min_quality = (int32_t)rs->var_quality[e0];
		for (uint32_t i = 1; i < cnt; ++i) min_quality = std::min(min_quality, (int32_t)rs->var_quality[e0 + i]);
		const int32_t good = (int32_t)cnt;
		const int32_t span = s.var_index[e0 + cnt - 1] - s.var_index[e0] + 1;
		const int32_t bad = good != span ? span - good : 0;
		s.initial_score[r] = Score{{good - bad, good - bad, min_quality}};
	}
	s.coverage.assign(s.n_variants, 0);
	s.selected.assign(s.n_reads, 0);

	if (n_preferred) {
		std::unordered_set<int32_t> ids(preferred_source_ids, preferred_source_ids + n_preferred);
		PySetInt preferred;
		for (uint32_t r = 0; r < s.n_reads; ++r) if (ids.count(read_source_id[r])) preferred.add(r);
		// The helper empties the set it is given, so the reference's `undecided_reads -= preferred_reads` that follows
		// (readselect.pyx:248) removes nothing: the second pass looks at every read again.
		if (preferred.size() > 0) s.helper(preferred);
	}
	PySetInt undecided = PySetInt::from_range(s.n_reads);
	s.helper(undecided);

	uint64_t n = 0;
	for (uint32_t r = 0; r < s.n_reads; ++r) { selected_out[r] = s.selected[r]; n += s.selected[r]; }
	if (n_selected) *n_selected = n;
	return WHAMD_OK;
}
