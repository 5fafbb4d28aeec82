"""The reference's OWN Python objects (ReadSet, Read, Pedigree, Genotype, PhredGenotypeLikelihoods from the compiled
whatshap.core) fed to the drop-in: flattening is identical to the mirror classes', and the oracle on the flattened
views returns what the reference's PedigreeDPTable returns for the same objects (CPU part; the device part is in
test_gpu_parity.py)."""

import numpy as np
import pytest

import oracle
from reference_cases import all_cases
from refobjects import reference_core, table_outputs, to_reference
from whatshap_amd.core import problem_from_objects

CASES = all_cases()


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_flattening_reference_objects_equals_mirror_objects(case):
    ref = reference_core()
    rs, ped = to_reference(case, ref)
    a = problem_from_objects(case.readset, case.recombcost, case.pedigree, case.distrust_genotypes, case.positions)
    b = problem_from_objects(rs, case.recombcost, ped.amd, case.distrust_genotypes, case.positions)
    for name in ("read_ptr", "var_position", "var_allele", "var_quality", "read_sample_id", "individual_id", "triple_ids",
                 "genotype", "recombcost"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    if a.genotype_likelihoods is None:
        assert b.genotype_likelihoods is None
    else:
        assert np.array_equal(a.genotype_likelihoods, b.genotype_likelihoods, equal_nan=True)


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_oracle_on_flattened_views_equals_reference_class(case):
    ref = reference_core()
    rs, ped = to_reference(case, ref)
    want = table_outputs(ref.PedigreeDPTable(rs, case.recombcost, ped, case.distrust_genotypes, case.positions))
    problem = problem_from_objects(rs, case.recombcost, ped.amd, case.distrust_genotypes, case.positions)
    table = oracle.OracleTable(problem)
    a0, a1, q, tv, sid = table.super_reads()
    positions = table.positions().tolist()
    got_reads = []
    for i in range(len(sid)):
        for h, alleles in ((0, a0), (1, a1)):
            got_reads.append((f"superread_{h}_{i}", int(sid[i]), -1, (-1,),
                              [(positions[c], int(alleles[i][c]), int(q[i][c])) for c in range(len(positions))]))
    assert int(table.optimal_score()) == want["cost"]
    assert table.partitioning().tolist() == want["partitioning"]
    assert tv.tolist() == want["transmission"]
    assert got_reads == want["superreads"]
