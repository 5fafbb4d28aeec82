"""Turns a reference_cases.Case (mirror classes) into the REFERENCE's own objects (whatshap.core, built into
oracle/_ref/cy by oracle/build_cython_ref.py) -- test infrastructure for the drop-in tests."""

import pytest

from oracle import build_cython_ref
from whatshap_amd import shim


def reference_core():
    if not build_cython_ref.available():
        pytest.skip("reference extension modules not built (oracle/_ref/cy)")
    return build_cython_ref.import_reference()


def to_reference(case, ref):
    """(reference ReadSet, recording reference Pedigree) holding exactly the case's data."""
    rs = ref.ReadSet()
    for read in case.readset:
        r = ref.Read(read.name, read.mapqs[0], read.source_id, read.sample_id)
        for v in read:
            r.add_variant(v.position, v.allele, v.quality)
        rs.add(r)
    mirror = case.pedigree
    names = {numeric: name for name, numeric in mirror.numeric_sample_ids.mapping.items()}
    pedigree_class = shim.recording_pedigree_class(ref.Pedigree)
    ids = ref.NumericSampleIds()  # the reference Pedigree only accepts its own class (core.pyx:420-422)
    for numeric in sorted(names):
        assert ids[names[numeric]] == numeric
    ped = pedigree_class(ids)
    for i, numeric in enumerate(mirror._ids):
        gts = [ref.Genotype(g.as_vector()) for g in mirror._genotypes[i]]
        gls = mirror._gls[i]
        ref_gls = None
        if any(g is not None for g in gls):
            ref_gls = [None if g is None else ref.PhredGenotypeLikelihoods(g.as_vector()) for g in gls]
        ped.add_individual(names[numeric], gts, ref_gls)
    for f, m, c in mirror._triples:
        ped.add_relationship(names[f], names[m], names[c])
    return rs, ped


def table_outputs(table):
    """Everything the PhasingAlgorithm interface returns, as plain Python data."""
    superreads, transmission = table.get_super_reads()
    reads = []
    for readset in superreads:
        for read in readset:
            reads.append((read.name, read.sample_id, read.source_id, tuple(read.mapqs),
                          [(v.position, v.allele, v.quality) for v in read]))
    return {"cost": int(table.get_optimal_cost()), "partitioning": [int(x) for x in table.get_optimal_partitioning()],
            "superreads": reads, "transmission": [int(t) for t in transmission]}


def problem_to_reference(problem, ref):
    """Reference ReadSet + recording Pedigree from flat arrays (sample names are the numeric ids as strings)."""
    rs = ref.ReadSet()
    ptr = problem.read_ptr
    for r in range(problem.n_reads):
        read = ref.Read(f"read{r}", 60, 0, int(problem.read_sample_id[r]))
        for i in range(int(ptr[r]), int(ptr[r + 1])):
            read.add_variant(int(problem.var_position[i]), int(problem.var_allele[i]), int(problem.var_quality[i]))
        rs.add(read)
    ids = ref.NumericSampleIds()
    individual_ids = [int(x) for x in problem.individual_id]
    for numeric in range(max(individual_ids) + 1):
        assert ids[str(numeric)] == numeric
    ped = shim.recording_pedigree_class(ref.Pedigree)(ids)
    n_ind, n_var = problem.n_individuals, problem.n_variants
    geno = problem.genotype.reshape(n_ind, n_var)
    gl = None if problem.genotype_likelihoods is None else problem.genotype_likelihoods.reshape(n_ind, n_var, 3)
    codes = {0: [0, 0], 1: [0, 1], 2: [1, 1]}
    for i, numeric in enumerate(individual_ids):
        gts = [ref.Genotype(codes[int(g)]) for g in geno[i]]
        gls = None if gl is None else [ref.PhredGenotypeLikelihoods([float(x) for x in gl[i, v]]) for v in range(n_var)]
        ped.add_individual(str(numeric), gts, gls)
    triples = problem.triple_ids.reshape(-1, 3)
    for f, m, c in triples:
        ped.add_relationship(str(int(f)), str(int(m)), str(int(c)))
    return rs, ped
