// whamd_ingest_helpers.h -- compiled ingestion of WhatsHap's own C++ objects (ReadSet, Pedigree) into the flat views of
// include/whatshap_amd.h.  Compiled against the reference's headers (src/readset.h, src/read.h, src/pedigree.h -- shipped in
// its sdist, MANIFEST.in:7-8,14) the way whatshap/readselect.pyx reaches the same objects (readselect.pyx:14-15,244); the
// non-inline members resolve against whatshap.core's shared object at load time (whatshap/__init__.py:6-18 imports it
// RTLD_GLOBAL for exactly that purpose).  Replaces the per-variant Python loop of whatshap_amd/core.py for reference objects.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>
#include <sys/mman.h>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "pedigree.h"
#include "read.h"
#include "readset.h"

// Large output arrays on huge pages: a fresh 33 MB of 4 KB pages is 8 000 page faults (20 ms in a container, and they serialise across
// the worker threads on the address-space lock); with MADV_HUGEPAGE it is 17.  (csrc/host_parallel.h allocates the planner's rows the same way.)
static inline void* whamd_ingest_alloc(size_t bytes) {
	const size_t huge = (size_t)2 << 20;
	if (bytes < huge) return std::malloc(bytes ? bytes : 1);
	const size_t rounded = (bytes + huge - 1) / huge * huge;
	void* ptr = nullptr;
	if (posix_memalign(&ptr, huge, rounded) != 0) return nullptr;
	if (ptr && getenv("WHAMD_NO_HUGEPAGES") == nullptr) (void)madvise(ptr, rounded, MADV_HUGEPAGE);
	return ptr;
}

// ---- worker threads for the walks below (reads / individuals x variants are independent): a few std::threads, none for small inputs
template <class F>
static inline void whamd_ingest_parallel(size_t n, size_t grain, F&& body) {
	size_t want = n / (grain ? grain : 1);
	unsigned hw = std::thread::hardware_concurrency();
	size_t threads = std::min<size_t>(std::min<size_t>(want, hw ? hw : 1), 8);
	if (const char* e = getenv("WHAMD_INGEST_THREADS")) threads = std::max(1, atoi(e));
	if (threads <= 1) { body((size_t)0, n); return; }
	std::vector<std::thread> pool;
	std::exception_ptr failure;
	std::mutex lock;
	for (size_t t = 0; t < threads; ++t) {
		const size_t lo = n * t / threads, hi = n * (t + 1) / threads;
		pool.emplace_back([&, lo, hi] {
			try { body(lo, hi); } catch (...) { std::lock_guard<std::mutex> g(lock); if (!failure) failure = std::current_exception(); }
		});
	}
	for (auto& th : pool) th.join();
	if (failure) std::rethrow_exception(failure);
}

// First pass over a ReadSet: the Read pointers and read_ptr[size + 1] (prefix sums of the variant counts).  Every Read is its own heap
// object -- 100 000 dependent cache misses when walked by one thread -- so the pass runs on the worker threads too; returns the variant count.
static inline size_t whamd_readset_scan(ReadSet* rs, std::vector<Read*>& ptr, uint64_t* read_ptr) {
	const size_t reads = (size_t)rs->size();
	ptr.resize(reads);
	whamd_ingest_parallel(reads, 4096, [&](size_t lo, size_t hi) {
		for (size_t r = lo; r < hi; ++r) {
			ptr[r] = rs->get((int)r);
			read_ptr[r + 1] = (uint64_t)ptr[r]->getVariantCount();
		}
	});
	read_ptr[0] = 0;
	for (size_t r = 0; r < reads; ++r) read_ptr[r + 1] += read_ptr[r];
	return (size_t)read_ptr[reads];
}

// What Read keeps per variant (src/read.h:53-57: `struct enriched_entry_t { Entry entry; int position; }` in ONE std::vector, Entry =
// {unsigned read_id; allele_t allele_type; unsigned phred_score}, src/entry.h:22-24).  Read::getEntry(0) (public, src/read.h:33) points at
// the first element's Entry, so the whole read is one contiguous array of these.  The view is CHECKED per read against the public getters
// (first and last variant) before it is used; a read where it does not hold is walked through the getters -- 3 calls into core.so per variant.
struct whamd_raw_variant { uint32_t read_id; int32_t allele; uint32_t phred; int32_t position; };

static inline bool whamd_raw_view_holds(const Read* read, const whamd_raw_variant* raw, int i) {
	return raw[i].position == read->getPosition(i) && raw[i].allele == read->getAllele(i) && (int)raw[i].phred == read->getVariantQuality(i);
}

// position / allele / quality [variant count], sample[size]  (whamd_readset_view)
// (ptr / read_ptr: what whamd_readset_scan left)
static inline void whamd_flatten_readset(const std::vector<Read*>& ptr, const uint64_t* read_ptr, int32_t* position, uint8_t* allele, uint32_t* quality, int32_t* sample) {
	static_assert(sizeof(whamd_raw_variant) == 16, "layout of Read's per-variant record");
	const size_t reads = ptr.size();
	whamd_ingest_parallel((size_t)reads, 2048, [&](size_t lo, size_t hi) {
		for (size_t r = lo; r < hi; ++r) {
			const Read* read = ptr[r];
			const int nv = (int)(read_ptr[r + 1] - read_ptr[r]);
			uint64_t at = read_ptr[r];
			sample[r] = (int32_t)read->getSampleID();
			if (nv == 0) continue;
			const whamd_raw_variant* raw = reinterpret_cast<const whamd_raw_variant*>(read->getEntry(0));
			const bool fast = sizeof(Entry) == 12 && whamd_raw_view_holds(read, raw, 0) && whamd_raw_view_holds(read, raw, nv - 1);
			for (int i = 0; i < nv; ++i, ++at) {
				const int a = fast ? raw[i].allele : read->getAllele(i);
				if (a < 0 || a > 255) throw std::runtime_error("read allele must be 0 (REF), 1 (ALT) or 2 (BLANK)");
				position[at] = fast ? raw[i].position : (int32_t)read->getPosition(i);
				allele[at] = (uint8_t)a;
				quality[at] = fast ? raw[i].phred : (uint32_t)read->getVariantQuality(i);
			}
		}
	});
}

// ids[size], triple_ids[3 * triple_count] (father, mother, child ids), genotype[size * variants] (WHAMD genotype codes),
// gl[size * variants * 3] (NaN where an individual has no likelihoods); returns 1 if any likelihood is present
static inline int whamd_flatten_pedigree(Pedigree* ped, uint32_t* ids, uint32_t* triple_ids, uint8_t* genotype, double* gl) {
	const size_t n_ind = ped->size(), n_var = n_ind ? ped->get_variant_count() : 0;
	std::atomic<int> any_gl(0);
	for (size_t i = 0; i < n_ind; ++i) ids[i] = ped->index_to_id(i);
	whamd_ingest_parallel(n_ind * n_var, 16384, [&](size_t lo, size_t hi) {
		int seen = 0;
		for (size_t at = lo; at < hi; ++at) {
			const size_t i = at / n_var, v = at % n_var;
			const Genotype* g = ped->get_genotype(i, v);
			uint8_t code = 255;   // WHAMD_GT_OTHER
			if (g != nullptr) {
				// Genotype::get_code() (src/genotype.h:118): ploidy in bits 60..63, allele j in bits 4j..4j+3 -- what is_diploid_and_biallelic()
				// and as_vector() (src/genotype.cpp:69-76,124-134) decode, without the vector they allocate
				const uint64_t word = g->get_code();
				const uint32_t a0 = (uint32_t)(word & 15u), a1 = (uint32_t)((word >> 4) & 15u);
				if (((word >> 60) & 15u) == 2 && a0 <= 1 && a1 <= 1) code = (uint8_t)(a0 + a1);
			}
			genotype[at] = code;
			const PhredGenotypeLikelihoods* l = ped->get_genotype_likelihoods(i, v);
			double* out = gl + at * 3;
			if (l != nullptr) {
				const std::vector<double>& values = l->as_vector();
				if (values.size() != 3) throw std::runtime_error("only diploid bi-allelic genotype likelihoods are supported");
				out[0] = values[0]; out[1] = values[1]; out[2] = values[2];
				seen = 1;
			} else {
				out[0] = out[1] = out[2] = std::numeric_limits<double>::quiet_NaN();
			}
		}
		if (seen) any_gl.store(1);
	});
	const std::vector<Pedigree::triple_entry_t>& triples = ped->get_triples();
	for (size_t t = 0; t < triples.size(); ++t)
		for (int m = 0; m < 3; ++m) triple_ids[3 * t + m] = ped->index_to_id(triples[t][m]);
	return any_gl.load();
}

// One individual's two superreads as a fresh reference ReadSet -- what PedigreeDPTable::get_super_reads builds per individual
// (src/pedigreedptable.cpp:354-387: Read("superread_<h>_<i>", mapq -1, source id -1, sample id = numeric id), one variant per
// column in column order, both haplotypes carry the SAME quality; the output set holds [superread_0, superread_1]).  The caller
// adopts the returned object exactly as whatshap/core.pyx:388-400 adopts the reference's.
// `numbered` == 0: the names of PedMecHeuristic::getSuperReads, "superread_0" / "superread_1" (src/pedmecheuristic.cpp:105-121).
static inline ReadSet* whamd_emit_superread_set(unsigned individual, int numbered, int sample_id, size_t n, const uint32_t* positions,
                                                const uint8_t* allele0, const uint8_t* allele1, const uint32_t* quality) {
	const std::string suffix = numbered ? "_" + std::to_string(individual) : std::string();
	Read* r0 = new Read("superread_0" + suffix, -1, -1, sample_id);
	Read* r1 = new Read("superread_1" + suffix, -1, -1, sample_id);
	// the two reads are independent objects: one worker each for long tables (Read::addVariant appends to the read's own vector)
	auto fill = [&](Read* r, const uint8_t* allele) {
		for (size_t c = 0; c < n; ++c) r->addVariant((int)positions[c], (int)allele[c], (int)quality[c]);
	};
	if (n >= 32768 && getenv("WHAMD_INGEST_THREADS") == nullptr) {
		std::thread other([&] { fill(r1, allele1); });
		fill(r0, allele0);
		other.join();
	} else {
		fill(r0, allele0);
		fill(r1, allele1);
	}
	ReadSet* out = new ReadSet();
	out->add(r0);
	out->add(r1);
	return out;
}

// All individuals at once, each on its own worker (sets[i] receives individual i's ReadSet).
static inline void whamd_emit_superread_sets(size_t n_ind, int numbered, const int32_t* sample_ids, size_t n, const uint32_t* positions,
                                             const uint8_t* allele0, const uint8_t* allele1, const uint32_t* quality, ReadSet** sets) {
	const size_t stride = n ? n : 1;   // (rows of the [individuals, columns] arrays; one padded element for an empty table)
	auto one = [&](size_t i) {
		sets[i] = whamd_emit_superread_set((unsigned)i, numbered, (int)sample_ids[i], n, positions, allele0 + i * stride, allele1 + i * stride, quality + i * stride);
	};
	if (n_ind > 1 && n >= 32768 && getenv("WHAMD_INGEST_THREADS") == nullptr) {
		std::vector<std::thread> pool;
		for (size_t i = 1; i < n_ind; ++i) pool.emplace_back(one, i);
		one(0);
		for (auto& th : pool) th.join();
	} else {
		for (size_t i = 0; i < n_ind; ++i) one(i);
	}
}

// source id of every read (whatshap/readselect.pyx:52-56 reads them one by one through Read.source_id)
static inline void whamd_read_source_ids(ReadSet* rs, int32_t* out) {
	whamd_ingest_parallel((size_t)rs->size(), 4096, [&](size_t lo, size_t hi) {
		for (size_t r = lo; r < hi; ++r) out[r] = (int32_t)rs->get((int)r)->getSourceID();
	});
}
