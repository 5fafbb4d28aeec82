// problem.cpp -- host precompute for the device DP and host post-processing of its result.
// See problem.h for the map to the reference's classes.
#include "debug_build.h"
#include "problem.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>

namespace whamd {

namespace {

// Pedigree::id_to_index (src/pedigree.cpp:47-55): later insertions of the same id win.
bool id_to_index(const Problem& p, uint32_t id, uint32_t& index, std::string& msg) {
	for (uint32_t i = p.n_ind; i-- > 0;) {
		if (p.individual_id[i] == id) {
			index = i;
			return true;
		}
	}
	msg = "Individual with ID " + std::to_string(id) + " not present in pedigree.";
	return false;
}

// PedigreePartitions (src/pedigreepartitions.cpp:7-42): founders get partitions (2r, 2r+1) in
// individual order; a child's haplotype 0 is the father's haplotype !(t >> 2*trio & 1), haplotype 1
// the mother's haplotype !(t >> (2*trio+1) & 1).
bool build_partitions(Problem& p, std::string& msg) {
	p.P = 2 * (p.n_ind - p.n_triples);
	p.h2p.assign((size_t)p.T * p.n_ind * 2, -1);
	std::vector<int> child_triple(p.n_ind, -1);
	for (uint32_t i = 0; i < p.n_triples; ++i) child_triple[p.triples[i][2]] = (int)i;
	// resolution order: parents before children (the reference recurses; a cycle would recurse forever there)
	std::vector<uint32_t> order;
	std::vector<uint8_t> state(p.n_ind, 0);
	for (uint32_t root = 0; root < p.n_ind; ++root) {
		std::vector<std::pair<uint32_t, int>> stack{{root, 0}};
		while (!stack.empty()) {
			auto [i, phase] = stack.back();
			stack.pop_back();
			if (state[i] == 2) continue;
			if (phase == 1) {
				state[i] = 2;
				order.push_back(i);
				continue;
			}
			if (state[i] == 1) {
				msg = "pedigree relationships contain a cycle";
				return false;
			}
			state[i] = 1;
			stack.push_back({i, 1});
			if (child_triple[i] >= 0) {
				stack.push_back({p.triples[child_triple[i]][1], 0});
				stack.push_back({p.triples[child_triple[i]][0], 0});
			}
		}
	}
	for (uint32_t t = 0; t < p.T; ++t) {
		int8_t* map = p.h2p.data() + (size_t)t * p.n_ind * 2;
		int next = 0;
		for (uint32_t i = 0; i < p.n_ind; ++i) {
			if (child_triple[i] < 0) {
				map[2 * i] = (int8_t)next;
				map[2 * i + 1] = (int8_t)(next + 1);
				next += 2;
			}
		}
		for (uint32_t i : order) {
			int ti = child_triple[i];
			if (ti < 0) continue;
			uint32_t father = p.triples[ti][0], mother = p.triples[ti][1];
			map[2 * i] = map[2 * father + (((t >> (2 * ti)) & 1) ? 0 : 1)];
			map[2 * i + 1] = map[2 * mother + (((t >> (2 * ti + 1)) & 1) ? 0 : 1)];
		}
	}
	return true;
}

}  // namespace

whamd_status_t build_problem(const whamd_readset_view* rs, const uint32_t* recombcost, size_t n_recombcost,
                             const whamd_pedigree_view* ped, bool distrust, const uint32_t* positions,
                             size_t n_positions, Problem& p, std::string& msg, bool columns_only, bool lazy_fact_terms) {
	if (!rs || !ped) {
		msg = "null readset or pedigree view";
		return WHAMD_ERR_INVALID;
	}
	// ---- copy the views
	const bool timing = getenv("WHAMD_DEBUG_TIMING") != nullptr;
	auto lap_t = std::chrono::steady_clock::now();
	long lap_faults = timing ? thread_minor_faults() : 0;
	auto lap = [&](const char* what) {
		if (!timing) return;
		const auto now = std::chrono::steady_clock::now();
		const long faults = thread_minor_faults();
		fprintf(stderr, "[whamd timing]   flatten: %s %.2f ms (%ld page faults of this thread)\n", what, std::chrono::duration<double, std::milli>(now - lap_t).count(), faults - lap_faults);
		lap_faults = faults;
		lap_t = now;
	};
	p.n_reads = rs->n_reads;
	const uint64_t nnz = rs->n_reads ? rs->read_ptr[rs->n_reads] : 0;
	p.read_ptr.assign(rs->n_reads + 1, 0);
	if (rs->n_reads) std::copy(rs->read_ptr, rs->read_ptr + rs->n_reads + 1, p.read_ptr.begin());
	// (the per-variant arrays of the view are read where they lie: the entries are the table's own copy of everything it needs later -- Problem::var_* stay empty;
	//  copying 7 MB per coverage-15 table was 0.3 of a create's 10 thread-ms)
	{
		// Entry::BLANK (2) inside a read is accepted and skipped by the reference (pedigreecolumncostcomputer.cpp:69-70,
		// 93-94), exactly like the BLANK entries ColumnIterator inserts; only EQUAL_SCORES (3) and beyond reach its
		// assert(false) (:71-72, asserts are live in the reference's build)
		const uint32_t n_threads = host_threads(nnz, 1u << 18);
		std::vector<uint8_t> bad(n_threads, 0);
		parallel_ranges(nnz, n_threads, [&](uint64_t i0, uint64_t i1, uint32_t t) {
			if (i1 <= i0) return;
			uint8_t worst = 0;
			for (uint64_t i = i0; i < i1; ++i) worst = std::max(worst, rs->var_allele[i]);
			bad[t] = worst > WHAMD_ALLELE_BLANK;
		});
		for (uint8_t v : bad) {
			if (v) {
				msg = "read allele must be 0 (REF), 1 (ALT) or 2 (BLANK)";
				return WHAMD_ERR_INVALID;
			}
		}
	}
	lap("copies of the read arrays");
	p.n_ind = ped->n_individuals;
	p.n_triples = ped->n_triples;
	p.n_variants = ped->n_variants;
	p.individual_id.assign(ped->individual_id, ped->individual_id + p.n_ind);
	const size_t ng = (size_t)p.n_ind * p.n_variants;
	if (ng) p.genotype.assign(ped->genotype, ped->genotype + ng);
	p.have_gl = ped->genotype_likelihoods != nullptr;
	if (p.have_gl && ng) {
		p.gl.assign(ped->genotype_likelihoods, ped->genotype_likelihoods + ng * 3);
		if (ped->gl_present) {
			for (size_t i = 0; i < ng; ++i) {
				if (!ped->gl_present[i]) p.gl[3 * i] = p.gl[3 * i + 1] = p.gl[3 * i + 2] = std::nan("");
			}
		}
	}
	p.distrust = distrust;
	p.triples.resize(p.n_triples);
	for (uint32_t i = 0; i < p.n_triples; ++i) {
		for (int m = 0; m < 3; ++m) {
			if (!id_to_index(p, ped->triple_ids[3 * i + m], p.triples[i][m], msg)) return WHAMD_ERR_INVALID;
		}
	}
	if (columns_only) {   // genotyping (genotype_device.hip)
		if (p.n_triples > 2 || p.n_ind > (uint32_t)MAX_IND) {
			msg = "pedigree too large for the genotyping device path (at most " + std::to_string(MAX_IND) + " individuals and 2 trios)";
			return WHAMD_ERR_UNSUPPORTED;
		}
	} else if (p.n_triples > (uint32_t)MAX_TRIPLES_WIDE || p.n_ind > (uint32_t)MAX_IND_WIDE || 2 * (p.n_ind - std::min(p.n_ind, p.n_triples)) > 14) {
		// (the reference has no such limit, src/pedigreepartitions.cpp:7-42; 4^trios transmission values x 2^partitions allele
		// assignments per column are enumerated on the host, src/pedigreecolumncostcomputer.cpp:25-49)
		msg = "pedigree too large for the device path (at most " + std::to_string(MAX_IND_WIDE) + " individuals, " + std::to_string(MAX_TRIPLES_WIDE) +
		      " trios, 14 founder haplotypes)";
		return WHAMD_ERR_UNSUPPORTED;
	}
	p.T = 1u << (2 * p.n_triples);

	// ---- ColumnIterator::ColumnIterator (src/columniterator.cpp:10-59): positions + validation
	if (positions == nullptr) {  // ReadSet::get_positions (src/readset.cpp:54-62)
		p.positions.resize(nnz);
		for (uint64_t i = 0; i < nnz; ++i) p.positions[i] = (uint32_t)rs->var_position[i];
		std::sort(p.positions.begin(), p.positions.end());
		p.positions.erase(std::unique(p.positions.begin(), p.positions.end()), p.positions.end());
	} else {
		p.positions.assign(positions, positions + n_positions);
	}
	p.n_cols = (uint32_t)p.positions.size();
	const uint32_t n = p.n_cols;
	lap("pedigree + positions");
	// positions -> columns: binary search in the (strictly increasing) position list; a list that is not -- rejected below, after
	// the reads, as before -- goes through the map the reference builds (later duplicates win, as position_map[pos] = i does)
	bool positions_increase = true;
	for (uint32_t i = 1; i < n && positions_increase; ++i) positions_increase = p.positions[i - 1] < p.positions[i];
	std::unordered_map<uint32_t, uint32_t> position_map;
	if (!positions_increase) {
		position_map.reserve(n * 2 + 1);
		for (uint32_t i = 0; i < n; ++i) position_map[p.positions[i]] = i;
	}
	// (hint: a column at or before the answer -- the previous read's first column, this read's first column: the search gallops from there
	// instead of bisecting the whole list, two cache misses instead of sixteen)
	auto column_of = [&](uint32_t position, uint32_t& col, uint32_t hint = 0) -> bool {
		if (positions_increase) {
			if (n == 0) return false;
			const uint32_t* pos0 = p.positions.data();
			size_t lo_i = hint < n ? hint : 0, step = 1;
			if (pos0[lo_i] > position) lo_i = 0;   // (a hint beyond the answer: fall back to the whole list)
			size_t hi_i = lo_i;
			while (hi_i < n && pos0[hi_i] < position) { lo_i = hi_i; hi_i += step; step <<= 1; }
			if (hi_i > n) hi_i = n;
			const auto it = std::lower_bound(p.positions.begin() + lo_i, p.positions.begin() + hi_i + (hi_i < n ? 1 : 0), position);
			if (it == p.positions.end() || *it != position) return false;
			col = (uint32_t)(it - p.positions.begin());
			return true;
		}
		const auto it = position_map.find(position);
		if (it == position_map.end()) return false;
		col = it->second;
		return true;
	};
	p.read_first_col.resize(p.n_reads);
	p.read_last_col.resize(p.n_reads);
	std::vector<uint32_t>&first_col = p.read_first_col, &last_col = p.read_last_col;
	{
		// the reads are independent but for the order test against the previous read's first position; the first failing read
		// (lowest index) decides the error, as in the sequential scan of the constructor
		struct ReadError { uint32_t read = 0xFFFFFFFFu; whamd_status_t status = WHAMD_OK; std::string msg; };
		const uint32_t n_threads = positions_increase ? host_threads(p.n_reads, 1u << 14) : 1u;
		std::vector<ReadError> errors(n_threads);
		const uint64_t* ptr = rs->read_ptr;
		const int32_t* vp = rs->var_position;
		parallel_ranges(p.n_reads, n_threads, [&](uint64_t r0, uint64_t r1, uint32_t t) {
			auto fail_read = [&](uint32_t r, whamd_status_t st, std::string m) { errors[t].read = r; errors[t].status = st; errors[t].msg = std::move(m); };
			uint32_t hint_col = 0;
			for (uint32_t r = (uint32_t)r0; r < (uint32_t)r1; ++r) {
				const uint64_t lo = ptr[r], hi = ptr[r + 1];
				if (hi <= lo || hi > nnz) return fail_read(r, WHAMD_ERR_INVALID, "No variants present");  // Read::firstPosition (src/read.cpp:75-78)
				// (an earlier read without variants fails first: this read's predecessor can be taken as valid)
				const int pos = r == 0 ? 0 : ((ptr[r] > ptr[r - 1] && ptr[r] <= nnz) ? vp[ptr[r - 1]] : 0);
				if (vp[lo] < pos) return fail_read(r, WHAMD_ERR_UNSORTED, "ColumnIterator: reads in ReadSet are not sorted.");
				for (uint64_t i = lo + 1; i < hi; ++i) {  // Read::isSorted (src/read.cpp:210-218): strictly increasing
					if (!(vp[i - 1] < vp[i])) return fail_read(r, WHAMD_ERR_UNSORTED, "ColumnIterator: encountered read with unsorted variants.");
				}
				uint32_t fc = 0, lc = 0;
				if (vp[lo] < 0 || !column_of((uint32_t)vp[lo], fc, hint_col) || !column_of((uint32_t)vp[hi - 1], lc, fc) || fc > lc) {
					// the reference asserts here (src/columniterator.cpp:36-39) and aborts the process
					return fail_read(r, WHAMD_ERR_INVALID, "read " + std::to_string(r) + " starts or ends at a position that is not in the position list");
				}
				first_col[r] = fc;
				last_col[r] = lc;
				hint_col = fc;   // (the reads are sorted by their first position)
			}
		});
		for (const ReadError& e : errors) {   // ranges are in read order
			if (e.status != WHAMD_OK) {
				msg = e.msg;
				return e.status;
			}
		}
	}
	if (!positions_increase) {
		msg = "positions must be strictly increasing";
		return WHAMD_ERR_INVALID;
	}
	lap("position map + read validation");
	// read -> individual (src/pedigreedptable.cpp:32-34)
	p.read_source.resize(p.n_reads);
	for (uint32_t r = 0; r < p.n_reads; ++r) {
		if (!id_to_index(p, (uint32_t)rs->read_sample_id[r], p.read_source[r], msg)) return WHAMD_ERR_INVALID;
	}
	if (n && p.n_ind && p.n_variants < n) {
		msg = "pedigree has fewer variants (" + std::to_string(p.n_variants) + ") than there are columns (" + std::to_string(n) + ")";
		return WHAMD_ERR_INVALID;
	}
	if (n && p.n_ind && distrust) {
		bool ok = p.have_gl;
		for (uint32_t i = 0; ok && i < p.n_ind; ++i) {
			for (uint32_t c = 0; ok && c < n; ++c) {
				for (int g = 0; g < 3; ++g) {
					const double v = p.gl[((size_t)i * p.n_variants + c) * 3 + g];
					if (!(v >= 0.0 && v <= 1e9)) ok = false;  // also rejects NaN (= missing GL; the reference asserts gls != nullptr)
				}
			}
		}
		if (!ok) {
			msg = "distrust_genotypes requires genotype likelihoods in [0, 1e9] for every individual and column";
			return WHAMD_ERR_INVALID;
		}
	}
	// recombination costs, padded (see include/whatshap_amd.h)
	p.recomb.assign(n, n_recombcost ? recombcost[n_recombcost - 1] : 0u);
	for (uint32_t c = 0; c < n && c < n_recombcost; ++c) p.recomb[c] = recombcost[c];

	if (!build_partitions(p, msg)) return WHAMD_ERR_INVALID;
	if (n == 0) return WHAMD_OK;

	// ---- columns (ColumnIterator::get_next, src/columniterator.cpp:91-139): read r is active in columns first_col[r]..last_col[r];
	// entries in read-index order; BLANK where the read has no variant.  The reads are sorted by their first position, so the reads
	// that start at column c are a contiguous range [start_idx[c], start_idx[c + 1]) and the coverage is a difference array.
	std::vector<uint32_t> start_idx(n + 2, 0);
	p.col_ptr.assign(n + 1, 0);
	{
		std::vector<int64_t> diff(n + 1, 0);
		for (uint32_t r = 0; r < p.n_reads; ++r) {
			start_idx[first_col[r] + 1]++;
			diff[first_col[r]]++;
			diff[last_col[r] + 1]--;
		}
		for (uint32_t c = 0; c <= n; ++c) start_idx[c + 1] += start_idx[c];
		int64_t cov = 0;
		for (uint32_t c = 0; c < n; ++c) {
			cov += diff[c];
			if (cov > (int64_t)MAX_COVERAGE) {
				msg = "coverage " + std::to_string(cov) + " at column " + std::to_string(c) + " exceeds the device limit of " + std::to_string(MAX_COVERAGE);
				return WHAMD_ERR_UNSUPPORTED;
			}
			p.col_ptr[c + 1] = p.col_ptr[c] + (uint64_t)cov;
		}
	}
	lap("column pointers");
	// ---- one pass per range of columns (a few host threads): the column's entries, ColumnIndexingScheme (k, backward width b, forward
	// mask / width f: src/columnindexingscheme.cpp:19-33 is a merge of two sorted id lists -- here the reads that continue are known
	// from their last column), then the per-bit deltas and the cost terms.  Each range keeps its own term list (concatenated in column
	// order afterwards; the sums are sums of integer-valued doubles, exact in any order).
	p.entries.resize(p.col_ptr[n]);
	p.k.resize(n);
	p.b.resize(n);
	p.f.resize(n);
	p.fwd_mask.resize(n);
	if (!columns_only) {
		p.delta.resize((size_t)p.col_ptr[n] * std::max<uint32_t>(p.n_ind, 1));
		p.term_ptr.assign((size_t)n * p.T + 1, 0);
		p.terms.clear();
	}
	struct RangeResult {
		std::vector<CostTerm> terms;
		double bound = 0.0;
		uint64_t n_cells = 0, algorithmic_bytes = 0;
		uint32_t max_k = 0;
		bool conflict = false;
		bool fact_ok = true;
	};
	// ---- factorised forms (Problem::fterms).  With untrusted genotypes every one of the 16 allele assignments (a, a', b, b') of a trio is
	// allowed: the child's haplotypes carry a (from founder X) and b (from founder Y), a' and b' are the alleles the founders did not
	// transmit, and the cost is g_X[a + a'] + g_Y[b + b'] + g_C[a + b] + cost_X + cost_Y + cost_C with, per individual and haplotype alleles
	// (h0, h1): (0,0) R, (1,1) W - R, (0,1) R + L, (1,0) W - R - L.  The minimum over a' and b' can be taken first:
	//   min_{a,b} [ g_C[a+b] + cost_C(a,b) + min_{a'} (g_X[a+a'] + cost_X) + min_{b'} (g_Y[b+b'] + cost_Y) ].
	// Which founder transmits to which haplotype of the child, and whether the transmitted allele sits on the founder's haplotype 0,
	// is read off the haplotype-to-partition map of the transmission value (FactRoles).  The construction is CHECKED, not trusted: for
	// every (column, transmission value, allele assignment) the constant and the L-dependence it implies must equal the term the generic
	// loop below computes; one difference and the table keeps the sixteen forms.
	struct FactRoles {
		int child = -1, X = -1, Y = -1;
		bool oX = false, oY = false;              // the transmitted allele sits on the founder's haplotype 0
		uint32_t bx = 0, bx2 = 0, by = 0, by2 = 0;   // allele bits: transmitted by X / X's other / transmitted by Y / Y's other
		bool ok = false;
	};
	bool want_fact = distrust && p.n_ind == 3 && p.P == 4 && p.T == 4 && !columns_only && !getenv("WHAMD_NO_PED_FACT");
	FactRoles roles[4];
	for (uint32_t t = 0; t < 4 && want_fact; ++t) {
		const int8_t* map = p.h2p.data() + (size_t)t * p.n_ind * 2;
		const int8_t* map0 = p.h2p.data();
		FactRoles r;
		int n_children = 0;
		for (uint32_t s = 0; s < 3; ++s) {   // the child: the individual whose haplotypes change partition with the transmission value
			bool varies = false;
			for (uint32_t tt = 1; tt < p.T; ++tt) {
				const int8_t* mt = p.h2p.data() + (size_t)tt * p.n_ind * 2;
				varies = varies || mt[2 * s] != map0[2 * s] || mt[2 * s + 1] != map0[2 * s + 1];
			}
			if (varies) { r.child = (int)s; ++n_children; }
		}
		int nx = 0, ny = 0;
		for (uint32_t s = 0; s < 3 && n_children == 1; ++s) {
			if ((int)s == r.child) continue;
			for (int h = 0; h < 2; ++h) {
				if (map[2 * s + h] == map[2 * r.child]) { r.X = (int)s; r.oX = h == 0; r.bx = (uint32_t)map[2 * s + h]; r.bx2 = (uint32_t)map[2 * s + 1 - h]; ++nx; }
				if (map[2 * s + h] == map[2 * r.child + 1]) { r.Y = (int)s; r.oY = h == 0; r.by = (uint32_t)map[2 * s + h]; r.by2 = (uint32_t)map[2 * s + 1 - h]; ++ny; }
			}
		}
		r.ok = n_children == 1 && nx == 1 && ny == 1 && r.X != r.Y;
		if (!r.ok) want_fact = false;
		roles[t] = r;
	}
	if (want_fact) { p.fterms.resize((size_t)n * p.T * 16); p.fterm_kind = 1; }   // (every entry is written below)
	// ---- a quartet (two children of the same two founders, T = 16): the same idea in HAPLOTYPE space (slots.h PSLOT_FACT4).  With (a0, a1) the alleles on X's
	// haplotypes and (b0, b1) on Y's, child k carries (a_uk, b_vk) where (u_k, v_k) is read off the haplotype-to-partition map of the transmission value:
	//   cost = g_X[a0 + a1] + cost_X(a0, a1) + g_Y[b0 + b1] + cost_Y(b0, b1) + sum over the children of g_Ck[a_uk + b_vk] + cost_Ck(a_uk, b_vk).
	// Nothing but the wiring depends on the value: ONE line of twenty entries per column.  Checked like the trio's: every (column, value, assignment) against the generic term.
	struct Fact4Roles {
		int X = -1, Y = -1, C1 = -1, C2 = -1;
		uint32_t px[2] = {0, 0}, py[2] = {0, 0};   // partitions of X's and Y's haplotypes
		uint32_t u[2][16], v[2][16];               // per child and transmission value
		bool ok = false;
	} roles4;
	bool want_fact4 = distrust && p.n_ind == 4 && p.P == 4 && p.T == 16 && p.n_triples == 2 && !columns_only && !getenv("WHAMD_NO_PED_FACT");
	if (want_fact4) {
		Fact4Roles& r = roles4;
		const int8_t* map0 = p.h2p.data();
		int n_children = 0;
		for (uint32_t s = 0; s < 4; ++s) {
			bool varies = false;
			for (uint32_t tt = 1; tt < p.T; ++tt) {
				const int8_t* mt = p.h2p.data() + (size_t)tt * p.n_ind * 2;
				varies = varies || mt[2 * s] != map0[2 * s] || mt[2 * s + 1] != map0[2 * s + 1];
			}
			if (varies) { (n_children == 0 ? r.C1 : r.C2) = (int)s; ++n_children; }
		}
		r.ok = n_children == 2;
		for (uint32_t s = 0; s < 4 && r.ok; ++s) {   // X transmits to haplotype 0 of child 1, Y to its haplotype 1
			if ((int)s == r.C1 || (int)s == r.C2) continue;
			for (int h = 0; h < 2; ++h) {
				if (map0[2 * s + h] == map0[2 * r.C1]) r.X = (int)s;
				if (map0[2 * s + h] == map0[2 * r.C1 + 1]) r.Y = (int)s;
			}
		}
		r.ok = r.ok && r.X >= 0 && r.Y >= 0 && r.X != r.Y;
		if (r.ok) {
			for (int h = 0; h < 2; ++h) { r.px[h] = (uint32_t)map0[2 * r.X + h]; r.py[h] = (uint32_t)map0[2 * r.Y + h]; }
			for (uint32_t t = 0; t < 16 && r.ok; ++t) {
				const int8_t* mt = p.h2p.data() + (size_t)t * p.n_ind * 2;
				r.ok = r.ok && (uint32_t)mt[2 * r.X] == r.px[0] && (uint32_t)mt[2 * r.X + 1] == r.px[1] && (uint32_t)mt[2 * r.Y] == r.py[0] && (uint32_t)mt[2 * r.Y + 1] == r.py[1];
				for (int k = 0; k < 2 && r.ok; ++k) {
					const int c = k == 0 ? r.C1 : r.C2;
					const uint32_t h0 = (uint32_t)mt[2 * c], h1 = (uint32_t)mt[2 * c + 1];
					r.ok = (h0 == r.px[0] || h0 == r.px[1]) && (h1 == r.py[0] || h1 == r.py[1]);
					r.u[k][t] = h0 == r.px[1] ? 1u : 0u;
					r.v[k][t] = h1 == r.py[1] ? 1u : 0u;
				}
			}
		}
		want_fact4 = r.ok;
		if (want_fact4) {
			p.fterms.resize((size_t)n * 20);
			p.fterm_kind = 2;
			p.fact4_roles[0] = p.fact4_roles[1] = 0;
			for (uint32_t t = 0; t < 16; ++t) {
				p.fact4_roles[0] |= (r.u[0][t] << t) | (r.v[0][t] << (16 + t));
				p.fact4_roles[1] |= ((r.u[0][t] == r.u[1][t] ? 1u : 0u) << t) | ((r.v[0][t] == r.v[1][t] ? 1u : 0u) << (16 + t));
			}
		}
	}
	// the twenty entries of column c: the signed sums L_X, L_Y, L_C1, L_C2, then per individual the cost of carrying (h0, h1) = 00, 01 (+ L), 10 (- L), 11
	auto factorised_line4 = [&](uint32_t c, const std::vector<uint32_t>& R, const std::vector<uint32_t>& W) -> CostTerm* {
		const Fact4Roles& r = roles4;
		auto g = [&](int s, uint32_t k) { return (uint32_t)(0.0 + p.gl[((size_t)s * p.n_variants + c) * 3 + k]); };
		CostTerm* ft = p.fterms.data() + (size_t)c * 20;
		const int who[4] = {r.X, r.Y, r.C1, r.C2};
		for (int i = 0; i < 4; ++i) {
			const int s = who[i];
			ft[i] = CostTerm{0, 1u << s, 0};
			ft[4 + 4 * i + 0] = CostTerm{g(s, 0) + R[s], 0, 0};
			ft[4 + 4 * i + 1] = CostTerm{g(s, 1) + R[s], 0, 0};          // + L_s
			ft[4 + 4 * i + 2] = CostTerm{g(s, 1) + W[s] - R[s], 0, 0};   // - L_s
			ft[4 + 4 * i + 3] = CostTerm{g(s, 2) + W[s] - R[s], 0, 0};
		}
		return ft;
	};
	auto factorised_form4 = [&](const CostTerm* ft, uint32_t t, uint32_t asg) -> CostTerm {
		const Fact4Roles& r = roles4;
		const uint32_t a[2] = {(asg >> r.px[0]) & 1u, (asg >> r.px[1]) & 1u}, b[2] = {(asg >> r.py[0]) & 1u, (asg >> r.py[1]) & 1u};
		const uint32_t h[4][2] = {{a[0], a[1]}, {b[0], b[1]}, {a[r.u[0][t]], b[r.v[0][t]]}, {a[r.u[1][t]], b[r.v[1][t]]}};
		const int who[4] = {r.X, r.Y, r.C1, r.C2};
		CostTerm tm{0, 0, 0};
		for (int i = 0; i < 4; ++i) {
			tm.c += ft[4 + 4 * i + h[i][0] * 2 + h[i][1]].c;
			if (h[i][0] != h[i][1]) (h[i][0] == 0 ? tm.plus : tm.minus) |= 1u << who[i];
		}
		return tm;
	};
	// the sixteen entries of (c, t): three signed sums {X_L, Y_L, C_L, 0}, then kx[4], ky[4], cc[4]
	auto factorised_line = [&](uint32_t c, uint32_t t, const std::vector<uint32_t>& R, const std::vector<uint32_t>& W) -> CostTerm* {
		const FactRoles& r = roles[t];
		auto g = [&](int s, uint32_t k) { return (uint32_t)(0.0 + p.gl[((size_t)s * p.n_variants + c) * 3 + k]); };   // (`cost += gls->get(genotype)` on an unsigned)
		CostTerm* ft = p.fterms.data() + ((size_t)c * p.T + t) * 16;
		auto signed_sum = [](int s, bool positive) { return positive ? CostTerm{0, 1u << s, 0} : CostTerm{0, 0, 1u << s}; };
		ft[0] = signed_sum(r.X, r.oX);   // X_L = +-L_X: + when the transmitted allele is on X's haplotype 0
		ft[1] = signed_sum(r.Y, r.oY);
		ft[2] = signed_sum(r.child, true);
		ft[3] = CostTerm{0, 0, 0};
		auto founder = [&](int s, bool o, CostTerm* k) {
			k[0] = CostTerm{g(s, 0) + R[s], 0, 0};                       // a = 0, a' = 0
			k[1] = CostTerm{g(s, 1) + (o ? R[s] : W[s] - R[s]), 0, 0};    // a = 0, a' = 1: + X_L
			k[2] = CostTerm{g(s, 2) + W[s] - R[s], 0, 0};                // a = 1, a' = 1
			k[3] = CostTerm{g(s, 1) + (o ? W[s] - R[s] : R[s]), 0, 0};    // a = 1, a' = 0: - X_L
		};
		founder(r.X, r.oX, ft + 4);
		founder(r.Y, r.oY, ft + 8);
		ft[12] = CostTerm{g(r.child, 0) + R[r.child], 0, 0};                   // (a, b) = (0, 0)
		ft[13] = CostTerm{g(r.child, 1) + R[r.child], 0, 0};                   // (0, 1): + L_child
		ft[14] = CostTerm{g(r.child, 1) + W[r.child] - R[r.child], 0, 0};      // (1, 0): - L_child
		ft[15] = CostTerm{g(r.child, 2) + W[r.child] - R[r.child], 0, 0};      // (1, 1)
		return ft;
	};
	// what the line implies for allele assignment `asg` (bits by partition, as the generic loop enumerates them)
	auto factorised_form = [&](const CostTerm* ft, uint32_t t, uint32_t asg) -> CostTerm {
		const FactRoles& r = roles[t];
		const uint32_t a = (asg >> r.bx) & 1u, a2 = (asg >> r.bx2) & 1u, b = (asg >> r.by) & 1u, b2 = (asg >> r.by2) & 1u;
		CostTerm tm{0, 0, 0};
		tm.c = ft[4 + (a == 0 ? (a2 == 0 ? 0 : 1) : (a2 == 1 ? 2 : 3))].c + ft[8 + (b == 0 ? (b2 == 0 ? 0 : 1) : (b2 == 1 ? 2 : 3))].c + ft[12 + a * 2 + b].c;
		if (a != a2) ((a == 0) == r.oX ? tm.plus : tm.minus) |= 1u << r.X;   // (a, a') = (0, 1): + X_L, which is + L_X when oX
		if (b != b2) ((b == 0) == r.oY ? tm.plus : tm.minus) |= 1u << r.Y;
		if (a != b) (a == 0 ? tm.plus : tm.minus) |= 1u << r.child;
		return tm;
	};
	// lazy generic terms (see column_terms): only where a factorised line exists
	bool lazy = lazy_fact_terms && (want_fact || want_fact4) && !debug_env("WHAMD_EAGER_TERMS");   // (switched off below if a sampled check fails)
	auto lazy_sample = [](uint32_t c) { return c < 256u || (c & 63u) == 0u; };
	p.lazy_terms = false;
	// (h2p of one individual without a trio: haplotype 0 -> partition 0, haplotype 1 -> partition 1)
	const bool single_trusted = p.n_ind == 1 && p.T == 1 && p.P == 2 && !distrust && p.h2p.size() >= 2 && p.h2p[0] == 0 && p.h2p[1] == 1 && !debug_env("WHAMD_NO_SINGLE_FAST_TERMS");
	// deltas and cost terms of column c (its entries and indexing scheme are in place); false: Mendelian conflict
	struct CompatCache {   // per worker: compatible allele assignments by (genotype vector, transmission value)
		bool enabled = false;
		std::vector<uint8_t> known;
		std::vector<std::vector<uint8_t>> lists;
	};
	auto make_compat = [&]() {
		CompatCache cc;
		cc.enabled = !distrust && p.n_ind >= 2 && p.n_ind <= 6 && p.P <= 8 && !debug_env("WHAMD_NO_COMPAT_CACHE");
		if (cc.enabled) {
			const size_t slots = ((size_t)1 << (2 * p.n_ind)) * p.T;
			cc.known.assign(slots, 0);
			cc.lists.resize(slots);
		}
		return cc;
	};
	auto column_terms = [&](uint32_t c, RangeResult& out, std::vector<uint32_t>& R, std::vector<uint32_t>& W, CompatCache& compat) -> bool {
		const ColumnEntry* col = p.col_begin(c);
		const uint32_t kc = p.k[c];
		std::fill(R.begin(), R.end(), 0u);
		std::fill(W.begin(), W.end(), 0u);
		int32_t* dl = p.delta.data() + (size_t)p.col_ptr[c] * p.n_ind;
		double wsum = 0.0;
		if (p.n_ind == 1) {
			// (one individual: every delta of the column is written, and REF / ALT is a coin flip per entry -- selects instead of branches: a mispredicted branch per
			//  entry was a quarter of this pass)
			uint32_t w = 0, r = 0;
			uint64_t wide = 0;
			for (uint32_t j = 0; j < kc; ++j) {
				const ColumnEntry& e = col[j];
				const uint32_t q = e.allele == WHAMD_ALLELE_BLANK ? 0u : e.phred;
				const uint32_t alt = e.allele == WHAMD_ALLELE_ALT ? q : 0u;
				w += q;
				wide += q;
				r += alt;
				dl[j] = (int32_t)(q - 2u * alt);   // REF +q, ALT -q, BLANK 0
			}
			W[0] = w;
			R[0] = r;
			wsum = (double)wide;
		} else {
		std::fill(dl, dl + (size_t)kc * p.n_ind, 0);
		uint64_t wide = 0;
		for (uint32_t j = 0; j < kc; ++j) {   // (the same selects; an entry writes the delta of ITS individual's row, the others stay 0)
			const ColumnEntry& e = col[j];
			const uint32_t q = e.allele == WHAMD_ALLELE_BLANK ? 0u : e.phred;
			const uint32_t alt = e.allele == WHAMD_ALLELE_ALT ? q : 0u;
			W[e.sample] += q;
			wide += q;
			R[e.sample] += alt;
			dl[(size_t)e.sample * kc + j] = (int32_t)(q - 2u * alt);
		}
		wsum = (double)wide;
		}
		double max_acost = 0.0;
		bool any = false;
		if (single_trusted) {
			// ONE individual, genotypes trusted (every table of `whatshap phase` without a pedigree): the enumeration below visits a = 0 .. 3 with
			// (a0, a1) = (a & 1, a >> 1) and keeps the assignments whose allele count equals the genotype -- written out: 0/0 -> {R}, 1/1 -> {W - R},
			// 0/1 -> a = 1: (1, 0) = W - R - L, then a = 2: (0, 1) = R + L (the same terms in the same order, src/pedigreecolumncostcomputer.cpp:25-49)
			const uint8_t g = p.genotype[c];
			if (g == 1) {
				out.terms.push_back(CostTerm{W[0] - R[0], 0, 1u});
				out.terms.push_back(CostTerm{R[0], 1u, 0});
				any = true;
			} else if (g == 0) {
				out.terms.push_back(CostTerm{R[0], 0, 0});
				any = true;
			} else if (g == 2) {
				out.terms.push_back(CostTerm{W[0] - R[0], 0, 0});
				any = true;
			}
			p.term_ptr[(size_t)c + 1] = out.terms.size();
		} else if (lazy && out.fact_ok && !lazy_sample(c)) {
			// A table whose runs read the FACTORISED line (Problem::fterms): the generic term list of this column -- sixteen terms per transmission value, 3 KB per column
			// of a quartet -- is read by nobody unless the column ends up outside every run, which the planner decides later: fill_lazy_terms builds those.  The
			// line itself is written here; its check against the generic enumeration runs on the sampled columns (the first 256 and every 64th: the identity is
			// algebraic in R, W and the likelihoods, what can be wrong is the role assignment, which is per table) and on every column built later.
			for (uint32_t t = 0; t < p.T; ++t) {
				if (want_fact) (void)factorised_line(c, t, R, W);
				else if (t == 0) (void)factorised_line4(c, R, W);
				p.term_ptr[(size_t)c * p.T + t + 1] = out.terms.size();
			}
			for (uint32_t s2 = 0; s2 < p.n_ind; ++s2) {   // (an upper bound of the largest likelihood sum of an assignment: every individual's largest)
				const double* g3 = p.gl.data() + ((size_t)s2 * p.n_variants + c) * 3;
				max_acost += std::max(g3[0], std::max(g3[1], g3[2]));
			}
			any = true;   // (untrusted genotypes: every assignment is allowed, a column cannot be infeasible)
		} else
		for (uint32_t t = 0; t < p.T; ++t) {
			const int8_t* map = p.h2p.data() + (size_t)t * p.n_ind * 2;
			const size_t begin = out.terms.size();
			const CostTerm* fline = (want_fact && out.fact_ok) ? factorised_line(c, t, R, W) : nullptr;
			const CostTerm* fline4 = (want_fact4 && out.fact_ok) ? (t == 0 ? factorised_line4(c, R, W) : p.fterms.data() + (size_t)c * 20) : nullptr;
			// trusted genotypes: WHICH assignments are compatible depends on the column only through its genotype vector -- looked up (a thread-local table
			// filled on first use) instead of tested sixteen times per transmission value and column: a quartet's 1 024 inner steps per column were 15 us
			const std::vector<uint8_t>* shortlist = nullptr;
			if (!distrust && compat.enabled) {
				uint32_t key = 0;
				for (uint32_t s = 0; s < p.n_ind; ++s) key = key * 4u + std::min<uint32_t>(p.genotype[(size_t)s * p.n_variants + c], 3u);
				const size_t slot = (size_t)key * p.T + t;
				if (!compat.known[slot]) {
					std::vector<uint8_t>& list = compat.lists[slot];
					for (uint32_t a = 0; a < (1u << p.P); ++a) {
						bool ok = true;
						for (uint32_t s = 0; s < p.n_ind && ok; ++s)
							ok = p.genotype[(size_t)s * p.n_variants + c] == ((a >> map[2 * s]) & 1) + ((a >> map[2 * s + 1]) & 1);
						if (ok) list.push_back((uint8_t)a);
					}
					compat.known[slot] = 1;
				}
				shortlist = &compat.lists[slot];
			}
			const uint32_t n_try = shortlist ? (uint32_t)shortlist->size() : (1u << p.P);
			for (uint32_t ai = 0; ai < n_try; ++ai) {  // src/pedigreecolumncostcomputer.cpp:25-49 (ascending assignment order either way)
				const uint32_t a = shortlist ? (*shortlist)[ai] : ai;
				bool compatible = true;
				uint32_t acost = 0;
				CostTerm term{0, 0, 0};
				for (uint32_t s = 0; s < p.n_ind; ++s) {
					const uint32_t a0 = (a >> map[2 * s]) & 1, a1 = (a >> map[2 * s + 1]) & 1;
					const size_t gi = (size_t)s * p.n_variants + c;
					if (distrust) {
						acost = (uint32_t)((double)acost + p.gl[gi * 3 + a0 + a1]);  // `cost += gls->get(genotype)` on an unsigned, :37
					} else if (p.genotype[gi] != a0 + a1) {
						compatible = false;
						break;
					}
					// cost of individual s if haplotype 0 carries a0 and haplotype 1 carries a1:
					//   (0,0): R_s   (1,1): W_s - R_s   (0,1): R_s + L_s(x)   (1,0): W_s - R_s - L_s(x)
					if (a0 == 0 && a1 == 0) term.c += R[s];
					else if (a0 == 1 && a1 == 1) term.c += W[s] - R[s];
					else if (a0 == 0) { term.c += R[s]; term.plus |= 1u << s; }
					else { term.c += W[s] - R[s]; term.minus |= 1u << s; }
				}
				if (!compatible) continue;
				term.c += acost;
				max_acost = std::max(max_acost, (double)acost);
				if (fline) {
					const CostTerm pred = factorised_form(fline, t, a);
					if (pred.c != term.c || pred.plus != term.plus || pred.minus != term.minus) { out.fact_ok = false; fline = nullptr; }
				}
				if (fline4) {
					const CostTerm pred = factorised_form4(fline4, t, a);
					if (pred.c != term.c || pred.plus != term.plus || pred.minus != term.minus) { out.fact_ok = false; fline4 = nullptr; }
				}
				// a term with the same L-dependence and a constant that is not smaller can never be the strict minimum
				bool dominated = false;
				for (size_t q = begin; q < out.terms.size(); ++q) {
					if (out.terms[q].plus == term.plus && out.terms[q].minus == term.minus) {
						if (term.c < out.terms[q].c) out.terms[q].c = term.c;
						dominated = true;
						break;
					}
				}
				if (!dominated) out.terms.push_back(term);
			}
			if (out.terms.size() > begin) any = true;
			p.term_ptr[(size_t)c * p.T + t + 1] = out.terms.size();   // relative to the range; rebased below
		}
		if (!any) {  // every transmission value infeasible at every cell (src/pedigreedptable.cpp:301-303)
			out.conflict = true;
			return false;
		}
		out.bound += wsum + max_acost + 2.0 * p.n_triples * (double)p.recomb[c];
		const uint64_t Tl = p.T;
		out.n_cells += 1ull << kc;
		out.algorithmic_bytes += (c > 0 ? 4 * Tl * (1ull << p.b[c]) : 0) + (c + 1 < n ? 12 * Tl * (1ull << p.f[c]) : 0) + 12ull * kc;
		return true;
	};
	auto columns_range = [&](uint32_t c_begin, uint32_t c_end, RangeResult& out) {
		if (c_begin >= c_end) return;
		out.terms.reserve(columns_only ? 0 : (size_t)(c_end - c_begin) * p.T * (distrust ? (size_t)1 << p.P : 2));   // (untrusted genotypes: every allele assignment is a term -- a vector that grows by doubling copied 150 MB several times)
		std::vector<uint32_t> R(p.n_ind), W(p.n_ind);
		CompatCache compat = make_compat();
		// Read-major: every read that touches the range writes its entry into each of its columns, in read order -- the rank of a read in a
		// column is the number of earlier reads active there, a counter per column.  (Column-major -- a list of active reads with a cursor each,
		// compacted and walked per column -- was 35 cycles per entry of dependent loads; this is sequential reads and one store per entry.)
		const uint32_t width = c_end - c_begin;
		std::vector<uint8_t> cnt(width, 0);
		std::vector<uint32_t> masks(width, 0);
		for (uint32_t r = 0; r < start_idx[c_end]; ++r) {   // (sorted by first column: the reads that start before the range ends)
			const uint32_t last = last_col[r];
			if (last < c_begin) continue;
			const uint32_t first = first_col[r];
			uint64_t v = p.read_ptr[r];
			if (first < c_begin) {   // started before the range: its first variant at or after the range's first position
				const int32_t* lo = rs->var_position + p.read_ptr[r];
				const int32_t* hi = rs->var_position + p.read_ptr[r + 1];
				v = (uint64_t)(std::lower_bound(lo, hi, (int32_t)p.positions[c_begin]) - rs->var_position);
			}
			const uint8_t sample = (uint8_t)p.read_source[r];
			const uint32_t c_hi = std::min(last, c_end - 1u);
			for (uint32_t c = std::max(first, c_begin); c <= c_hi; ++c) {
				const int cpos = (int)p.positions[c];
				while (rs->var_position[v] < cpos) ++v;   // (the read's last variant lies in column `last`: v stays inside the read)
				const uint32_t j = cnt[c - c_begin]++;
				ColumnEntry& e = p.entries[p.col_ptr[c] + j];
				e.read_id = r;
				e.sample = sample;
				if (rs->var_position[v] == cpos) {
					e.allele = rs->var_allele[v];
					e.phred = rs->var_quality[v];
				} else {
					e.allele = WHAMD_ALLELE_BLANK;
					e.phred = 0;
				}
				if (last > c) masks[c - c_begin] |= 1u << j;
			}
		}
		for (uint32_t c = c_begin; c < c_end; ++c) {
			const uint32_t kc = cnt[c - c_begin], mask = masks[c - c_begin];
			p.k[c] = (uint8_t)kc;
			p.b[c] = (uint8_t)(kc - (start_idx[c + 1] - start_idx[c]));   // the reads that started earlier: the low bits of the column's index
			p.fwd_mask[c] = mask;   // last column: everything is minimised out (global optimum, src/pedigreedptable.cpp:306-315)
			p.f[c] = (uint8_t)__builtin_popcount(mask);
			out.max_k = std::max(out.max_k, kc);
			if (!columns_only && !column_terms(c, out, R, W, compat)) return;
		}
	};
	double bound = 0.0;  // upper bound on any DP value, to rule out 32-bit wrap-around
	{
		const uint32_t n_threads = host_threads(n, 8192);
		std::vector<RangeResult> parts(n_threads);
		std::vector<uint32_t> bounds(n_threads + 1);
		for (uint32_t t = 0; t <= n_threads; ++t) bounds[t] = (uint32_t)((uint64_t)n * t / n_threads);
		parallel_ranges(n, n_threads, [&](uint64_t c0, uint64_t c1, uint32_t t) { columns_range((uint32_t)c0, (uint32_t)c1, parts[t]); });
		if (lazy) {
			bool all_ok = true;
			for (uint32_t t = 0; t < n_threads; ++t) all_ok = all_ok && parts[t].fact_ok;
			if (!all_ok) {
				// a sampled column's line disagrees with the generic terms: the table keeps its generic forms -- for EVERY column, so the pass runs again without the shortcut
				lazy = false;
				for (RangeResult& part : parts) part = RangeResult();
				parallel_ranges(n, n_threads, [&](uint64_t c0, uint64_t c1, uint32_t t) { columns_range((uint32_t)c0, (uint32_t)c1, parts[t]); });
			}
		}
		p.lazy_terms = lazy;
		for (uint32_t t = 0; t < n_threads; ++t) p.max_k = std::max(p.max_k, parts[t].max_k);
		lap("column entries, indexing scheme, deltas + cost terms");
		if (columns_only) return WHAMD_OK;   // the genotyping path (genotype.cpp) has its own per-column model
		std::vector<uint64_t> base(n_threads + 1, 0);
		for (uint32_t t = 0; t < n_threads; ++t) {
			if (parts[t].conflict) {
				msg = "Error: Mendelian conflict";
				return WHAMD_ERR_MENDELIAN_CONFLICT;
			}
			base[t + 1] = base[t] + parts[t].terms.size();
			bound += parts[t].bound;
			if (!parts[t].fact_ok) { p.fterms.clear(); p.fterm_kind = 0; }
			p.n_cells += parts[t].n_cells;
			p.algorithmic_bytes += parts[t].algorithmic_bytes;
		}
		// the term lists of the ranges, concatenated in column order (each range copies its own piece: 72 MB for an untrusted trio of 100 000 columns)
		p.terms.resize(base[n_threads]);
		parallel_ranges(n_threads, n_threads, [&](uint64_t t0, uint64_t t1, uint32_t) {
			for (uint64_t t = t0; t < t1; ++t) {
				for (size_t i = (size_t)bounds[t] * p.T + 1; i <= (size_t)bounds[t + 1] * p.T; ++i) p.term_ptr[i] += base[t];
				if (!parts[t].terms.empty()) std::memcpy(p.terms.data() + base[t], parts[t].terms.data(), parts[t].terms.size() * sizeof(CostTerm));
			}
		});
	}
	lap("term lists joined");
	p.value_bound = bound;
	if (bound >= 4294967295.0) {
		msg = "costs may exceed 32 bits (upper bound " + std::to_string(bound) + "); the reference's unsigned arithmetic wraps there and results are undefined";
		return WHAMD_ERR_OVERFLOW;
	}
	return WHAMD_OK;
}

// The generic term lists of the columns in `need` that do not have them yet (Problem::lazy_terms: build_problem left them out where the runs read the
// factorised line).  The enumeration is the one of build_problem's column_terms (src/pedigreecolumncostcomputer.cpp:25-49: every allele assignment, its likelihood
// cost accumulated as `(u32)((double)cost + gl)`, terms with the same L-dependence merged into the cheaper one, in ascending assignment order).
whamd_status_t fill_lazy_terms(Problem& p, const std::vector<uint8_t>& need, std::string& msg) {
	if (!p.lazy_terms) return WHAMD_OK;
	const uint32_t n = p.n_cols;
	if (p.terms_built.size() != n) {   // first call: what build_problem built (the sampled columns)
		p.terms_built.assign(n, 0);
		for (uint32_t c = 0; c < n; ++c) p.terms_built[c] = p.term_end(c, p.T - 1) > p.term_begin(c, 0);
	}
	std::vector<uint32_t> todo;
	for (uint32_t c = 0; c < n; ++c) if (need[c] && !p.terms_built[c]) todo.push_back(c);
	if (todo.empty()) return WHAMD_OK;
	// per column its new terms, then one merge pass over all columns
	std::vector<std::vector<CostTerm>> fresh(todo.size());
	std::vector<std::vector<uint32_t>> fresh_count(todo.size());
	parallel_ranges(todo.size(), host_threads(todo.size(), 256), [&](uint64_t i0, uint64_t i1, uint32_t) {
		std::vector<uint32_t> R(p.n_ind), W(p.n_ind);
		for (size_t i = i0; i < i1; ++i) {
			const uint32_t c = todo[i];
			const ColumnEntry* col = p.col_begin(c);
			std::fill(R.begin(), R.end(), 0u);
			std::fill(W.begin(), W.end(), 0u);
			for (uint32_t j = 0; j < p.k[c]; ++j) {
				const ColumnEntry& e = col[j];
				if (e.allele == WHAMD_ALLELE_BLANK) continue;
				W[e.sample] += e.phred;
				if (e.allele == WHAMD_ALLELE_ALT) R[e.sample] += e.phred;
			}
			std::vector<CostTerm>& out = fresh[i];
			fresh_count[i].assign(p.T, 0);
			for (uint32_t t = 0; t < p.T; ++t) {
				const int8_t* map = p.h2p.data() + (size_t)t * p.n_ind * 2;
				const size_t begin = out.size();
				for (uint32_t a = 0; a < (1u << p.P); ++a) {
					bool compatible = true;
					uint32_t acost = 0;
					CostTerm term{0, 0, 0};
					for (uint32_t s = 0; s < p.n_ind; ++s) {
						const uint32_t a0 = (a >> map[2 * s]) & 1, a1 = (a >> map[2 * s + 1]) & 1;
						const size_t gi = (size_t)s * p.n_variants + c;
						if (p.distrust) acost = (uint32_t)((double)acost + p.gl[gi * 3 + a0 + a1]);
						else if (p.genotype[gi] != a0 + a1) { compatible = false; break; }
						if (a0 == 0 && a1 == 0) term.c += R[s];
						else if (a0 == 1 && a1 == 1) term.c += W[s] - R[s];
						else if (a0 == 0) { term.c += R[s]; term.plus |= 1u << s; }
						else { term.c += W[s] - R[s]; term.minus |= 1u << s; }
					}
					if (!compatible) continue;
					term.c += acost;
					bool dominated = false;
					for (size_t q = begin; q < out.size(); ++q) {
						if (out[q].plus == term.plus && out[q].minus == term.minus) {
							if (term.c < out[q].c) out[q].c = term.c;
							dominated = true;
							break;
						}
					}
					if (!dominated) out.push_back(term);
				}
				fresh_count[i][t] = (uint32_t)(out.size() - begin);
			}
		}
	});
	RawVec<CostTerm> merged;
	std::vector<uint64_t> ptr((size_t)n * p.T + 1, 0);
	size_t total = p.terms.size();
	for (const auto& v : fresh) total += v.size();
	merged.resize(total);
	size_t at = 0, ti = 0;
	for (uint32_t c = 0; c < n; ++c) {
		const bool is_new = ti < todo.size() && todo[ti] == c;
		size_t off = 0;
		for (uint32_t t = 0; t < p.T; ++t) {
			ptr[(size_t)c * p.T + t] = at;
			if (is_new) {
				const uint32_t cnt = fresh_count[ti][t];
				if (cnt) std::memcpy(merged.data() + at, fresh[ti].data() + off, cnt * sizeof(CostTerm));
				off += cnt;
				at += cnt;
			} else {
				const uint64_t b = p.term_begin(c, t), e = p.term_end(c, t);
				if (e > b) std::memcpy(merged.data() + at, p.terms.data() + b, (e - b) * sizeof(CostTerm));
				at += e - b;
			}
		}
		if (is_new) { p.terms_built[c] = 1; ++ti; }
	}
	ptr[(size_t)n * p.T] = at;
	merged.resize(at);
	p.terms.swap(merged);
	p.term_ptr.swap(ptr);
	(void)msg;
	return WHAMD_OK;
}

// get_super_reads / get_alleles / get_optimal_partitioning on the host from the finished path.
// Columns are independent given the path, so they are split over a few host threads (results per column are written
// to disjoint slots; the partition flag of a read is only ever set to 0, with a relaxed atomic store).
namespace {

whamd_status_t finish_columns(const Problem& p, Solution& s, uint32_t c_begin, uint32_t c_end, std::string& msg) {
	const uint32_t n = p.n_cols;
	if (p.n_ind == 1 && p.T == 1 && p.P == 2 && !p.distrust && p.h2p.size() >= 2 && p.h2p[0] == 0 && p.h2p[1] == 1 && !debug_env("WHAMD_GENERIC_FINISH")) {
		// ONE individual, genotypes trusted (every table of `whatshap phase` without a pedigree): the loops below written out.  Partition p = side of the read;
		// cp[p][1] += q for REF, cp[p][0] += q for ALT (set_partitioning, :53-76); the assignments compatible with genotype 0/1 are a = 1 (haplotype 0 carries ALT:
		// cost cp[0][1] + cp[1][0]) then a = 2 (cost cp[0][0] + cp[1][1]), `<=` lets the later one win a tie (:131); both haplotypes' quality is the absolute
		// difference of the two; a homozygous genotype has ONE assignment and the other allele's best stays INF = -1 as an int (:162): quality cost + 1.
		// (96 coverage-15 tables spent 600 thread-ms per step in the general loop: 5 ms per table, the tail of every shared solve.)
		for (uint32_t c = c_begin; c < c_end; ++c) {
			const uint32_t x = s.path_index[c];
			const ColumnEntry* col = p.col_begin(c);
			const uint32_t kc = p.k[c];
			// (no branch on the side or on the allele: both are coin flips per entry, and a mispredicted branch costs more than the whole rest of the
			//  iteration -- 3.3 ms per coverage-15 table of 50 000 columns with the branches.  acc[side][allele]: REF 0, ALT 1, BLANK 2 is added and ignored;
			//  the store of a read on side 1 goes to a scratch byte.)
			uint32_t acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
			uint8_t scratch;
			for (uint32_t j = 0; j < kc; ++j) {
				const ColumnEntry& e = col[j];
				const uint32_t side = (x >> j) & 1u;
				uint8_t* dst = side ? &scratch : &s.partition[e.read_id];
				__atomic_store_n(dst, (uint8_t)0, __ATOMIC_RELAXED);
				acc[side][e.allele & 3u] += e.phred;
			}
			const uint32_t cp[2][2] = {{acc[0][WHAMD_ALLELE_ALT], acc[0][WHAMD_ALLELE_REF]}, {acc[1][WHAMD_ALLELE_ALT], acc[1][WHAMD_ALLELE_REF]}};
			const uint8_t g = p.genotype[c];
			uint8_t a0, a1;
			uint32_t quality;
			if (g == 1) {
				const uint32_t cost1 = cp[0][1] + cp[1][0], cost2 = cp[0][0] + cp[1][1];
				const bool second = cost2 <= cost1;           // a = 2 is visited last
				if ((second ? cost2 : cost1) == INF) { msg = "Error: Mendelian conflict"; return WHAMD_ERR_MENDELIAN_CONFLICT; }
				a0 = second ? 0 : 1;
				a1 = second ? 1 : 0;
				quality = (uint32_t)std::abs((int)cost1 - (int)cost2);   // |best[h][0] - best[h][1]| for either haplotype
				if (quality == 0) a0 = a1 = WHAMD_ALLELE_EQUAL_SCORES;
			} else if (g == 0 || g == 2) {
				const uint32_t al = g == 2 ? 1u : 0u;
				const uint32_t cost = cp[0][al] + cp[1][al];
				if (cost == INF) { msg = "Error: Mendelian conflict"; return WHAMD_ERR_MENDELIAN_CONFLICT; }
				a0 = a1 = (uint8_t)al;
				quality = (uint32_t)std::abs((int)cost - (int)INF);     // the other allele was never feasible: INF reads as -1 (:162)
				if (quality == 0) a0 = a1 = WHAMD_ALLELE_EQUAL_SCORES;
			} else {
				msg = "Error: Mendelian conflict";
				return WHAMD_ERR_MENDELIAN_CONFLICT;
			}
			s.allele0[c] = a0;
			s.allele1[c] = a1;
			s.quality[c] = quality;
		}
		return WHAMD_OK;
	}
	std::vector<std::array<uint32_t, 2>> cp(std::max<uint32_t>(p.P, 1));
	std::vector<std::array<uint32_t, 4>> best_for(std::max<uint32_t>(p.n_ind, 1));  // [ind][hap*2 + allele]
	for (uint32_t c = c_begin; c < c_end; ++c) {
		const uint32_t x = s.path_index[c], t = s.path_trans[c];
		const ColumnEntry* col = p.col_begin(c);
		const uint32_t kc = p.k[c];
		const int8_t* map = p.h2p.data() + (size_t)t * p.n_ind * 2;
		// partitioning: read is in partition 0 wherever its bit is 0 (src/pedigreedptable.cpp:398-400, core.pyx:414)
		for (uint32_t j = 0; j < kc; ++j) {
			if (((x >> j) & 1u) == 0) __atomic_store_n(&s.partition[col[j].read_id], (uint8_t)0, __ATOMIC_RELAXED);
		}
		// set_partitioning (src/pedigreecolumncostcomputer.cpp:53-76)
		for (auto& v : cp) v = {0, 0};
		for (uint32_t j = 0; j < kc; ++j) {   // (selects, not branches: REF / ALT is a coin flip per entry)
			const ColumnEntry& e = col[j];
			const int part = map[2 * e.sample + ((x >> j) & 1u)];
			cp[part][1] += e.allele == WHAMD_ALLELE_REF ? e.phred : 0u;
			cp[part][0] += e.allele == WHAMD_ALLELE_ALT ? e.phred : 0u;
		}
		// get_alleles (src/pedigreecolumncostcomputer.cpp:117-175)
		uint32_t best = INF;
		for (auto& v : best_for) v = {INF, INF, INF, INF};
		uint32_t best_a = 0;
		bool have = false;
		for (uint32_t a = 0; a < (1u << p.P); ++a) {
			bool compatible = true;
			uint32_t cost = 0;
			for (uint32_t i = 0; i < p.n_ind; ++i) {
				const uint32_t a0 = (a >> map[2 * i]) & 1, a1 = (a >> map[2 * i + 1]) & 1;
				const size_t gi = (size_t)i * p.n_variants + c;
				if (p.distrust) cost = (uint32_t)((double)cost + p.gl[gi * 3 + a0 + a1]);
				else if (p.genotype[gi] != a0 + a1) { compatible = false; break; }
			}
			if (!compatible) continue;
			for (uint32_t q = 0; q < p.P; ++q) cost += cp[q][(a >> q) & 1];
			if (cost <= best) {  // `<=`: the last minimal assignment wins (:131)
				best = cost;
				best_a = a;
				have = true;
			}
			for (uint32_t i = 0; i < p.n_ind; ++i) {
				const uint32_t a0 = (a >> map[2 * i]) & 1, a1 = (a >> map[2 * i + 1]) & 1;
				best_for[i][a0] = std::min(best_for[i][a0], cost);
				best_for[i][2 + a1] = std::min(best_for[i][2 + a1], cost);
			}
		}
		if (!have || best == INF) {
			msg = "Error: Mendelian conflict";
			return WHAMD_ERR_MENDELIAN_CONFLICT;
		}
		for (uint32_t i = 0; i < p.n_ind; ++i) {
			uint8_t a0 = (uint8_t)((best_a >> map[2 * i]) & 1), a1 = (uint8_t)((best_a >> map[2 * i + 1]) & 1);
			const int q0 = std::abs((int)best_for[i][0] - (int)best_for[i][1]);
			const int q1 = std::abs((int)best_for[i][2] - (int)best_for[i][3]);
			if (q0 == 0) a0 = WHAMD_ALLELE_EQUAL_SCORES;
			if (q1 == 0) a1 = WHAMD_ALLELE_EQUAL_SCORES;
			s.allele0[(size_t)i * n + c] = a0;
			s.allele1[(size_t)i * n + c] = a1;
			s.quality[(size_t)i * n + c] = (uint32_t)q1;  // only the haplotype-1 value survives (:162-163)
		}
	}
	return WHAMD_OK;
}

}  // namespace

whamd_status_t finish_solution(const Problem& p, Solution& s, std::string& msg) {
	const uint32_t n = p.n_cols;
	if (s.superreads_done && p.n_ind == 1 && s.allele0.size() == n && s.allele1.size() == n && s.quality.size() == n) {
		// The device made the superreads; what is left is get_optimal_partitioning (src/pedigreedptable.cpp:391-406): a read is in partition 0 if its bit of the
		// index is 0.  The bit of a read is the same in every column it covers (the backtrace carries it from column to column through the projection), so the
		// column where the read enters decides: its entries are the last k - b of that column.
		s.partition.assign(p.n_reads, 1);
		for (uint32_t r = 0; r < p.n_reads; ++r) {
			const uint32_t c = p.read_first_col[r];
			if (c >= n) continue;
			const ColumnEntry* col = p.col_begin(c);
			const uint32_t kc = p.k[c];
			uint32_t j = std::min<uint32_t>(p.b[c], kc);
			while (j < kc && col[j].read_id != r) ++j;
			if (j == kc) for (j = 0; j < kc && col[j].read_id != r; ++j) {}
			if (j < kc) s.partition[r] = (uint8_t)((s.path_index[c] >> j) & 1u);
		}
		return WHAMD_OK;
	}
	s.allele0.assign((size_t)p.n_ind * n, 0);
	s.allele1.assign((size_t)p.n_ind * n, 0);
	s.quality.assign((size_t)p.n_ind * n, 0);
	s.partition.assign(p.n_reads, 1);
	const uint32_t n_threads = n < 20000 ? 1u : host_threads(n, 8192);
	if (n_threads == 1) return finish_columns(p, s, 0, n, msg);
	std::vector<whamd_status_t> status(n_threads, WHAMD_OK);
	std::vector<std::string> messages(n_threads);
	parallel_ranges(n, n_threads, [&](uint64_t c0, uint64_t c1, uint32_t w) { status[w] = finish_columns(p, s, (uint32_t)c0, (uint32_t)c1, messages[w]); });
	for (uint32_t w = 0; w < n_threads; ++w) {
		if (status[w] != WHAMD_OK) {
			msg = messages[w];
			return status[w];
		}
	}
	return WHAMD_OK;
}

}  // namespace whamd
