// genotype.cpp -- host side of the genotyping path: the per-column model (transition and prior tables, error
// probabilities) that genotype_device.hip evaluates.  See genotype.h.
#include "genotype.h"

#include <cmath>

namespace whamd {

whamd_status_t build_genotype_model(const Problem& p, GenotypeModel& m, std::string& msg) {
	const uint32_t n = p.n_cols, T = p.T, ni = p.n_ind;
	if (p.P > 4) {
		msg = "genotyping on the device supports up to 4 haplotype partitions (a trio or a quartet), this pedigree has " + std::to_string(p.P);
		return WHAMD_ERR_UNSUPPORTED;
	}
	m.A = 1u << p.P;
	const uint32_t A = m.A;
	if (n && ni == 0) {
		msg = "genotyping needs at least one individual";
		return WHAMD_ERR_INVALID;
	}
	// ---- error probabilities of the column entries (get_phred_probability, src/genotypecolumncostcomputer.cpp:26-48)
	m.error_prob.resize(p.entries.size());
	double small[256];   // (the reference keeps the same table, :27-35)
	small[0] = (double)0.9999L;
	for (int q = 1; q < 256; ++q) small[q] = (double)powl(10.0L, -(long double)q / 10.0L);
	for (size_t e = 0; e < p.entries.size(); ++e) {
		const uint32_t q = p.entries[e].phred;
		m.error_prob[e] = q < 256 ? small[q] : (double)powl(10.0L, -(long double)(int)q / 10.0L);
	}
	// ---- genotype of every individual under (transmission value, allele assignment)
	m.genotype_index.assign((size_t)T * A * std::max<uint32_t>(ni, 1), 0);
	for (uint32_t i = 0; i < T; ++i)
		for (uint32_t a = 0; a < A; ++a)
			for (uint32_t s = 0; s < ni; ++s) {
				const uint32_t p0 = (uint32_t)p.h2p[((size_t)i * ni + s) * 2 + 0], p1 = (uint32_t)p.h2p[((size_t)i * ni + s) * 2 + 1];
				m.genotype_index[((size_t)i * A + a) * ni + s] = (uint8_t)(((a >> p0) & 1u) + ((a >> p1) & 1u));
			}
	// ---- transitions between transmission values (src/transitionprobabilitycomputer.cpp:22-45)
	const uint32_t nb = 2 * p.n_triples + 1;
	m.transition_bern.assign((size_t)n * nb, 0.0);
	for (uint32_t c = 0; c < n; ++c) {
		const long double r = powl(10.0L, -(long double)p.recomb[c] / 10.0L);
		long double bern[5];
		for (uint32_t x = 0; x < nb; ++x) bern[x] = powl(r, (long double)x) * powl(1.0L - r, (long double)(2 * p.n_triples - x));
		long double norm = 0.0L;
		for (uint32_t j = 0; j < T; ++j) norm += bern[__builtin_popcount(j)];   // row 0; every row is a permutation of it
		for (uint32_t x = 0; x < nb; ++x) m.transition_bern[(size_t)c * nb + x] = (double)(bern[x] / norm);
	}
	// ---- priors of the allele assignments (:48-90)
	m.allele_prior.assign((size_t)n * T * A, 0.0);
	if (n) {
		if (!p.have_gl) {
			msg = "genotyping requires genotype likelihoods (priors) for every individual and column";
			return WHAMD_ERR_INVALID;
		}
		std::vector<long double> prior(A);
		std::vector<uint32_t> key(A);
		for (uint32_t c = 0; c < n; ++c) {
			for (uint32_t s = 0; s < ni; ++s)
				for (int g = 0; g < 3; ++g)
					if (std::isnan(p.gl[((size_t)s * p.n_variants + c) * 3 + g])) {
						msg = "genotyping requires genotype likelihoods (priors) for every individual and column";
						return WHAMD_ERR_INVALID;   // the reference asserts gls != nullptr (:66)
					}
			for (uint32_t i = 0; i < T; ++i) {
				uint32_t count[81] = {0};   // genotype vectors in base 3: at most 4 individuals (P <= 4)
				for (uint32_t a = 0; a < A; ++a) {
					long double pr = 1.0L;
					uint32_t k = 0;
					for (uint32_t s = 0; s < ni; ++s) {
						const uint32_t g = m.genotype_index[((size_t)i * A + a) * ni + s];
						pr *= p.gl[((size_t)s * p.n_variants + c) * 3 + g];
						k = k * 3 + g;
					}
					prior[a] = pr;
					key[a] = k;
					++count[k];
				}
				long double norm = 0.0L;
				for (uint32_t a = 0; a < A; ++a) { prior[a] /= (long double)count[key[a]]; norm += prior[a]; }
				for (uint32_t a = 0; a < A; ++a) m.allele_prior[((size_t)c * T + i) * A + a] = (double)(prior[a] / norm);
			}
		}
	}
	return WHAMD_OK;
}

}  // namespace whamd
