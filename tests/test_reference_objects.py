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


def test_shim_install_rebinds_a_phase_like_module():
    """whatshap_amd.shim.install on a stand-in for whatshap.cli.phase (which binds Pedigree / PedigreeDPTable at import,
    cli/phase.py:34-42): the pedigree class becomes the recording subclass of the reference's, the table class the device
    factory; the recorded pedigree equals what the mirror classes hold; the previous bindings are returned."""
    import types

    from whatshap_amd import shim
    from whatshap_amd.core import Pedigree as MirrorPedigree

    ref = reference_core()
    phase = types.SimpleNamespace(Pedigree=ref.Pedigree, PedigreeDPTable=ref.PedigreeDPTable)
    previous = shim.install(phase, ref)
    assert previous == (ref.Pedigree, ref.PedigreeDPTable)
    assert issubclass(phase.Pedigree, ref.Pedigree) and phase.Pedigree is not ref.Pedigree
    ids = ref.NumericSampleIds()
    ped = phase.Pedigree(ids)  # what create_pedigree() does (cli/phase.py:901-935)
    gts = [ref.Genotype([0, 1]), ref.Genotype([1, 1]), ref.Genotype([0, 0])]
    gls = [ref.PhredGenotypeLikelihoods([10, 0, 20]), None, ref.PhredGenotypeLikelihoods([0, 5, 7])]
    ped.add_individual("father", gts, gls)
    ped.add_individual("mother", gts)
    ped.add_individual("child", gts)
    ped.add_relationship("father", "mother", "child")
    assert len(ped) == 3 and ped.variant_count == 3  # the reference object is fully functional
    rec = ped.amd
    assert isinstance(rec, MirrorPedigree) and rec._ids == [0, 1, 2] and rec._triples == [(0, 1, 2)]
    assert [g.as_vector() for g in rec._genotypes[0]] == [[0, 1], [1, 1], [0, 0]]
    assert rec._gls[0][0].as_vector() == [10.0, 0.0, 20.0] and rec._gls[0][1] is None and rec._gls[1] == [None] * 3
    with pytest.raises(TypeError):
        phase.PedigreeDPTable(ref.ReadSet(), [1, 1, 1], ref.Pedigree(ids))  # a pedigree that was not recorded


def test_shim_falls_back_to_the_replaced_class_beyond_device_limits(monkeypatch):
    """shim.install must not turn inputs the reference can phase into hard errors.  A refusal of the device
    path for its OWN limits (UNSUPPORTED / OVERFLOW / DEVICE) is logged and the table is built by the class that was
    replaced, from the original objects; algorithmic errors are re-raised; without a fallback class the refusal stays
    an error (what tests and bench.py rely on)."""
    import types

    import pytest
    from whatshap_amd import _native, core, shim

    calls = []

    class FakeReferenceTable:
        def __init__(self, readset, recombcost, pedigree, distrust_genotypes=False, positions=None):
            calls.append((readset, tuple(recombcost), pedigree, distrust_genotypes, positions))

    def refusing(status, message):
        def ctor(*args, **kwargs):
            raise _native.SolverError(status, message)
        return ctor

    ped = core.Pedigree(core.NumericSampleIds())
    rs = core.ReadSet()
    phase = types.SimpleNamespace(Pedigree=core.Pedigree, PedigreeDPTable=FakeReferenceTable)
    previous = shim.install(phase, None)
    assert previous == (core.Pedigree, FakeReferenceTable)
    for status in (_native.WHAMD_ERR_UNSUPPORTED, _native.WHAMD_ERR_OVERFLOW, _native.WHAMD_ERR_DEVICE):
        monkeypatch.setattr(core, "PedigreeDPTable", refusing(status, "beyond the device path"))
        table = phase.PedigreeDPTable(rs, [1, 2], ped, True, [10, 20])
        assert isinstance(table, FakeReferenceTable) and calls[-1] == (rs, (1, 2), ped, True, [10, 20])
    monkeypatch.setattr(core, "PedigreeDPTable", refusing(_native.WHAMD_ERR_MENDELIAN_CONFLICT, "Error: Mendelian conflict"))
    with pytest.raises(RuntimeError, match="Mendelian conflict"):
        phase.PedigreeDPTable(rs, [1, 2], ped)
    monkeypatch.setattr(core, "PedigreeDPTable", refusing(_native.WHAMD_ERR_UNSUPPORTED, "beyond the device path"))
    with pytest.raises(_native.SolverError):
        shim.table_factory(None)(rs, [1, 2], ped)
