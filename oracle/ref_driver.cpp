/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A thin extern "C" driver around the REAL reference classes (ReadSet, Pedigree, PedigreeDPTable),
 * compiled together with the reference's own translation units where they lie under
 * /root/reference/src (see oracle/Makefile; output goes to oracle/_ref/, which is git-ignored).
 * No reference source is copied into this repository: this file only #includes the reference
 * headers and calls the public API that whatshap/core.pyx:364-416 calls.
 *
 * It consumes the same views as the product library (include/whatshap_amd.h), so the oracle
 * restatement (pedmec_oracle.c), the compiled reference and the HIP path can be compared on
 * byte-identical inputs.  It is also the "reference" CPU baseline of bench.py.
 */
#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

/* index_path / indexers are private members (src/pedigreedptable.h:26-54); the parity tests want the
 * raw backtrace as well, so open the class up for this one translation unit. */
#define private public
#include "pedigreedptable.h"
#undef private
#include "pedigree.h"
#include "readset.h"
#include "genotype.h"
#include "phredgenotypelikelihoods.h"
#include "pedmecheuristic.h"

#include "../include/whatshap_amd.h"

struct whref_table {
	ReadSet* readset = nullptr;
	Pedigree* pedigree = nullptr;
	std::vector<unsigned int>* positions = nullptr;
	std::vector<unsigned int> recombcost;
	PedigreeDPTable* table = nullptr;
	unsigned int n_cols = 0;
	double ctor_seconds = 0.0;
	std::string err;
};

extern "C" {

int whref_create(const whamd_readset_view* rs, const uint32_t* recombcost, size_t n_recombcost,
                 const whamd_pedigree_view* ped, int distrust_genotypes, const uint32_t* positions,
                 size_t n_positions, whref_table** out) {
	whref_table* t = new whref_table();
	*out = t;
	try {
		t->readset = new ReadSet();
		for (uint32_t r = 0; r < rs->n_reads; ++r) {
			Read* read = new Read("r" + std::to_string(r), 50, 0, rs->read_sample_id[r]);
			for (uint64_t i = rs->read_ptr[r]; i < rs->read_ptr[r + 1]; ++i)
				read->addVariant(rs->var_position[i], rs->var_allele[i], (int)rs->var_quality[i]);
			t->readset->add(read);
		}
		t->pedigree = new Pedigree();
		for (uint32_t i = 0; i < ped->n_individuals; ++i) {
			std::vector<Genotype*> gts;
			std::vector<PhredGenotypeLikelihoods*> gls;
			for (uint32_t v = 0; v < ped->n_variants; ++v) {
				size_t gi = (size_t)i * ped->n_variants + v;
				uint8_t g = ped->genotype[gi];
				if (g == 0) gts.push_back(new Genotype(std::vector<uint32_t>{0, 0}));
				else if (g == 1) gts.push_back(new Genotype(std::vector<uint32_t>{0, 1}));
				else if (g == 2) gts.push_back(new Genotype(std::vector<uint32_t>{1, 1}));
				else gts.push_back(new Genotype());
				if (ped->genotype_likelihoods && (!ped->gl_present || ped->gl_present[gi])) {
					const double* p = ped->genotype_likelihoods + gi * 3;
					gls.push_back(new PhredGenotypeLikelihoods(std::vector<double>{p[0], p[1], p[2]}, 2, 2));
				} else {
					gls.push_back(nullptr);
				}
			}
			t->pedigree->addIndividual(ped->individual_id[i], gts, gls);
		}
		for (uint32_t i = 0; i < ped->n_triples; ++i)
			t->pedigree->addRelationship(ped->triple_ids[3 * i], ped->triple_ids[3 * i + 1], ped->triple_ids[3 * i + 2]);
		if (positions) t->positions = new std::vector<unsigned int>(positions, positions + n_positions);
		t->recombcost.assign(recombcost, recombcost + n_recombcost);
		auto t0 = std::chrono::steady_clock::now();
		t->table = new PedigreeDPTable(t->readset, t->recombcost, t->pedigree, distrust_genotypes != 0, t->positions);
		auto t1 = std::chrono::steady_clock::now();
		t->ctor_seconds = std::chrono::duration<double>(t1 - t0).count();
		t->n_cols = t->table->input_column_iterator.get_column_count();
	} catch (const std::exception& e) {
		t->err = e.what();
		return 1;
	}
	return 0;
}

const char* whref_error(const whref_table* t) { return t->err.c_str(); }
uint32_t whref_column_count(const whref_table* t) { return t->n_cols; }
double whref_ctor_seconds(const whref_table* t) { return t->ctor_seconds; }
uint32_t whref_optimal_score(const whref_table* t) { return t->table->get_optimal_score(); }

void whref_positions(const whref_table* t, uint32_t* out) {
	const std::vector<unsigned int>* p = t->table->input_column_iterator.get_positions();
	for (size_t i = 0; i < p->size(); ++i) out[i] = p->at(i);
}

void whref_index_path(const whref_table* t, uint32_t* index_out, uint32_t* trans_out) {
	for (size_t i = 0; i < t->table->index_path.size(); ++i) {
		index_out[i] = t->table->index_path[i].index;
		trans_out[i] = t->table->index_path[i].inheritance_value;
	}
}

int whref_super_reads(whref_table* t, uint8_t* allele0, uint8_t* allele1, uint32_t* quality,
                      uint32_t* transmission, uint32_t* sample_id) {
	try {
		std::vector<ReadSet*> out;
		for (size_t i = 0; i < t->pedigree->size(); ++i) out.push_back(new ReadSet());
		std::vector<unsigned int> tv;
		t->table->get_super_reads(&out, &tv);
		size_t n = t->n_cols;
		for (size_t i = 0; i < out.size(); ++i) {
			Read* r0 = out[i]->get(0);
			Read* r1 = out[i]->get(1);
			sample_id[i] = (uint32_t)r0->getSampleID();
			for (size_t c = 0; c < (size_t)r0->getVariantCount(); ++c) {
				allele0[i * n + c] = (uint8_t)r0->getAllele(c);
				allele1[i * n + c] = (uint8_t)r1->getAllele(c);
				quality[i * n + c] = (uint32_t)r0->getVariantQuality(c);
				if (r0->getVariantQuality(c) != r1->getVariantQuality(c)) throw std::runtime_error("superread qualities differ");
			}
			delete out[i];
		}
		for (size_t c = 0; c < tv.size(); ++c) transmission[c] = tv[c];
	} catch (const std::exception& e) {
		t->err = e.what();
		return 1;
	}
	return 0;
}

void whref_partitioning(const whref_table* t, uint8_t* out) {
	std::vector<bool>* p = t->table->get_optimal_partitioning();
	for (size_t i = 0; i < p->size(); ++i) out[i] = (*p)[i] ? 0 : 1; /* core.pyx:414 */
	delete p;
}

void whref_destroy(whref_table* t) {
	if (!t) return;
	delete t->table;
	delete t->readset;
	delete t->pedigree;
	/* t->positions is leaked by the reference wrapper too (core.pyx:370-375); free it here */
	delete t->positions;
	delete t;
}

/* ---- PedMecHeuristic (src/pedmecheuristic.cpp; whatshap/core.pyx:674-734; selected at whatshap/cli/phase.py:589-603) ----
 * The sibling solver behind the same API: a beam of at most row_limit partial solutions per column, float scores.
 * The ReadSet must be sorted (the CLI calls all_reads.sort() first); the pedigree's individuals must be in ascending
 * sample-id order (PedMecHeuristic reads genotypes by the RANK of the sample id, src/pedmecheuristic.cpp:62-78). */
struct whref_heuristic {
	ReadSet* readset = nullptr;
	Pedigree* pedigree = nullptr;
	std::vector<unsigned int>* positions = nullptr;
	std::vector<unsigned int> recombcost;
	PedMecHeuristic* solver = nullptr;
	unsigned int n_cols = 0, n_samples = 0;
	double solve_seconds = 0.0;
	std::string err;
};

int whref_heuristic_create(const whamd_readset_view* rs, const uint32_t* recombcost, size_t n_recombcost,
                           const whamd_pedigree_view* ped, int distrust_genotypes, const uint32_t* positions,
                           size_t n_positions, uint32_t row_limit, int allow_mutations, whref_heuristic** out) {
	whref_heuristic* t = new whref_heuristic();
	*out = t;
	try {
		t->readset = new ReadSet();
		for (uint32_t r = 0; r < rs->n_reads; ++r) {
			Read* read = new Read("r" + std::to_string(r), 50, 0, rs->read_sample_id[r]);
			for (uint64_t i = rs->read_ptr[r]; i < rs->read_ptr[r + 1]; ++i)
				read->addVariant(rs->var_position[i], rs->var_allele[i], (int)rs->var_quality[i]);
			t->readset->add(read);
		}
		t->pedigree = new Pedigree();
		for (uint32_t i = 0; i < ped->n_individuals; ++i) {
			std::vector<Genotype*> gts;
			std::vector<PhredGenotypeLikelihoods*> gls;
			for (uint32_t v = 0; v < ped->n_variants; ++v) {
				const uint8_t g = ped->genotype[(size_t)i * ped->n_variants + v];
				gts.push_back(new Genotype(std::vector<uint32_t>{g >= 2 ? 1u : 0u, g >= 1 ? 1u : 0u}));
				gls.push_back(nullptr);
			}
			t->pedigree->addIndividual(ped->individual_id[i], gts, gls);
		}
		for (uint32_t i = 0; i < ped->n_triples; ++i)
			t->pedigree->addRelationship(ped->triple_ids[3 * i], ped->triple_ids[3 * i + 1], ped->triple_ids[3 * i + 2]);
		if (positions) t->positions = new std::vector<unsigned int>(positions, positions + n_positions);
		t->recombcost.assign(recombcost, recombcost + n_recombcost);
		t->solver = new PedMecHeuristic(t->readset, t->recombcost, t->pedigree, distrust_genotypes != 0, t->positions, row_limit, allow_mutations != 0, 0);
		auto t0 = std::chrono::steady_clock::now();
		t->solver->solve();
		t->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		std::vector<Transmission>* tv = t->solver->getOptTransmission();
		t->n_cols = (unsigned int)tv->size();
		delete tv;
		t->n_samples = (unsigned int)t->solver->getOptHaplotypes().size();
	} catch (const std::exception& e) {
		t->err = e.what();
		return 1;
	}
	return 0;
}

const char* whref_heuristic_error(const whref_heuristic* t) { return t->err.c_str(); }
uint32_t whref_heuristic_column_count(const whref_heuristic* t) { return t->n_cols; }
uint32_t whref_heuristic_sample_count(const whref_heuristic* t) { return t->n_samples; }
double whref_heuristic_solve_seconds(const whref_heuristic* t) { return t->solve_seconds; }
float whref_heuristic_score(const whref_heuristic* t) { return t->solver->getOptScore(); }

/* raw getOptBipartition() bits (core.pyx:711-717 returns 0 where the bit is set, 1 where it is not) */
void whref_heuristic_bipartition(const whref_heuristic* t, uint8_t* out) {
	Bipartition* b = t->solver->getOptBipartition();
	for (size_t i = 0; i < b->size(); ++i) out[i] = (*b)[i] ? 1 : 0;
	delete b;
}

void whref_heuristic_transmission(const whref_heuristic* t, uint32_t* out) {
	std::vector<Transmission>* tv = t->solver->getOptTransmission();
	for (size_t i = 0; i < tv->size(); ++i) out[i] = (*tv)[i];
	delete tv;
}

/* haplotypes[sample][hap][column] as int8 (-1 never occurs after solve()); mutated[sample][hap][column] as 0 / 1 */
void whref_heuristic_haplotypes(const whref_heuristic* t, int8_t* haps, uint8_t* mutated) {
	const std::vector<std::vector<std::vector<Allele>>> h = t->solver->getOptHaplotypes();
	const size_t n = t->n_cols;
	for (size_t s = 0; s < h.size(); ++s)
		for (size_t hap = 0; hap < 2; ++hap)
			for (size_t c = 0; c < n; ++c) haps[(s * 2 + hap) * n + c] = h[s][hap][c];
	std::memset(mutated, 0, h.size() * 2 * n);
	std::vector<std::vector<std::pair<uint32_t, uint32_t>>>* m = t->solver->getMutations();
	for (size_t s = 0; s < m->size(); ++s)
		for (const std::pair<uint32_t, uint32_t>& e : (*m)[s]) mutated[(s * 2 + e.first) * n + e.second] = 1;
	delete m;
}

void whref_heuristic_destroy(whref_heuristic* t) {
	if (!t) return;
	delete t->solver;
	delete t->readset;
	delete t->pedigree;
	delete t->positions;
	delete t;
}

} /* extern "C" */
