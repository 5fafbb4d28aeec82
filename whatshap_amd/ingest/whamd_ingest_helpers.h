// whamd_ingest_helpers.h -- compiled ingestion of WhatsHap's own C++ objects (ReadSet, Pedigree) into the flat views of
// include/whatshap_amd.h.  Compiled against the reference's headers (src/readset.h, src/read.h, src/pedigree.h -- shipped in
// its sdist, MANIFEST.in:7-8,14) the way whatshap/readselect.pyx reaches the same objects (readselect.pyx:14-15,244); the
// non-inline members resolve against whatshap.core's shared object at load time (whatshap/__init__.py:6-18 imports it
// RTLD_GLOBAL for exactly that purpose).  Replaces the per-variant Python loop of whatshap_amd/core.py for reference objects.
#pragma once
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "pedigree.h"
#include "read.h"
#include "readset.h"

static inline size_t whamd_readset_variant_count(ReadSet* rs) {
	size_t n = 0;
	const int reads = (int)rs->size();
	for (int r = 0; r < reads; ++r) n += (size_t)rs->get(r)->getVariantCount();
	return n;
}

// read_ptr[size + 1], position / allele / quality [variant count], sample[size]  (whamd_readset_view)
static inline void whamd_flatten_readset(ReadSet* rs, uint64_t* read_ptr, int32_t* position, uint8_t* allele, uint32_t* quality, int32_t* sample) {
	const int reads = (int)rs->size();
	uint64_t at = 0;
	read_ptr[0] = 0;
	for (int r = 0; r < reads; ++r) {
		Read* read = rs->get(r);
		const int nv = read->getVariantCount();
		for (int i = 0; i < nv; ++i, ++at) {
			position[at] = (int32_t)read->getPosition(i);
			const int a = read->getAllele(i);
			if (a < 0 || a > 255) throw std::runtime_error("read allele must be 0 (REF), 1 (ALT) or 2 (BLANK)");
			allele[at] = (uint8_t)a;
			quality[at] = (uint32_t)read->getVariantQuality(i);
		}
		read_ptr[r + 1] = at;
		sample[r] = (int32_t)read->getSampleID();
	}
}

// ids[size], triple_ids[3 * triple_count] (father, mother, child ids), genotype[size * variants] (WHAMD genotype codes),
// gl[size * variants * 3] (NaN where an individual has no likelihoods); returns 1 if any likelihood is present
static inline int whamd_flatten_pedigree(Pedigree* ped, uint32_t* ids, uint32_t* triple_ids, uint8_t* genotype, double* gl) {
	const size_t n_ind = ped->size(), n_var = ped->get_variant_count();
	int any_gl = 0;
	for (size_t i = 0; i < n_ind; ++i) {
		ids[i] = ped->index_to_id(i);
		for (size_t v = 0; v < n_var; ++v) {
			const Genotype* g = ped->get_genotype(i, v);
			uint8_t code = 255;   // WHAMD_GT_OTHER
			if (g != nullptr && g->is_diploid_and_biallelic()) {
				const std::vector<uint32_t> alleles = g->as_vector();
				code = (uint8_t)(alleles[0] + alleles[1]);
			}
			genotype[i * n_var + v] = code;
			const PhredGenotypeLikelihoods* l = ped->get_genotype_likelihoods(i, v);
			double* out = gl + (i * n_var + v) * 3;
			if (l != nullptr) {
				const std::vector<double> values = l->as_vector();
				if (values.size() != 3) throw std::runtime_error("only diploid bi-allelic genotype likelihoods are supported");
				out[0] = values[0]; out[1] = values[1]; out[2] = values[2];
				any_gl = 1;
			} else {
				out[0] = out[1] = out[2] = std::numeric_limits<double>::quiet_NaN();
			}
		}
	}
	const std::vector<Pedigree::triple_entry_t>& triples = ped->get_triples();
	for (size_t t = 0; t < triples.size(); ++t)
		for (int m = 0; m < 3; ++m) triple_ids[3 * t + m] = ped->index_to_id(triples[t][m]);
	return any_gl;
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
	for (size_t c = 0; c < n; ++c) {
		r0->addVariant((int)positions[c], (int)allele0[c], (int)quality[c]);
		r1->addVariant((int)positions[c], (int)allele1[c], (int)quality[c]);
	}
	ReadSet* out = new ReadSet();
	out->add(r0);
	out->add(r1);
	return out;
}

// source id of every read (whatshap/readselect.pyx:52-56 reads them one by one through Read.source_id)
static inline void whamd_read_source_ids(ReadSet* rs, int32_t* out) {
	const int reads = (int)rs->size();
	for (int r = 0; r < reads; ++r) out[r] = (int32_t)rs->get(r)->getSourceID();
}
