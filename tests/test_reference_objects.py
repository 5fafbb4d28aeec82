"""The reference's OWN Python objects (ReadSet, Read, Pedigree, Genotype, PhredGenotypeLikelihoods from the compiled
whatshap.core) fed to the drop-in: flattening is identical to the mirror classes', and the oracle on the flattened
views returns what the reference's PedigreeDPTable returns for the same objects (CPU part; the device part is in
test_gpu_parity.py)."""

import numpy as np
import pytest

import oracle
from reference_cases import all_cases
from oracle import build_cython_ref
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
    def reference_readselection(readset, max_cov, preferred_source_ids=None, bridging=True):   # stands for whatshap.readselect's
        raise AssertionError("the reference's readselection must not be called once the shim is installed")

    phase = types.SimpleNamespace(Pedigree=ref.Pedigree, PedigreeDPTable=ref.PedigreeDPTable, PedMecHeuristic=ref.PedMecHeuristic,
                                  readselection=reference_readselection)
    previous = shim.install(phase, ref)
    assert previous == (ref.Pedigree, ref.PedigreeDPTable)
    assert phase.PedMecHeuristic is not ref.PedMecHeuristic and previous.bindings["PedMecHeuristic"] is ref.PedMecHeuristic
    # cli/phase.py:43,163: `select_reads` calls the module-level name `readselection` -- rebound too, and it takes reference ReadSets
    assert phase.readselection is not reference_readselection and previous.bindings["readselection"] is reference_readselection
    rs = ref.ReadSet()
    for r in range(12):
        read = ref.Read(f"r{r}", 60, r % 2, 0)
        for i in range(3):
            read.add_variant(100 * (r // 2 + i + 1), (r + i) % 2, 10 + r)
        rs.add(read)
    rs.sort()
    shim.reset_stats()
    for preferred in (None, {1}):
        got = phase.readselection(rs, 3, preferred)
        assert isinstance(got, set) and got and got <= set(range(len(rs)))
        assert got == __import__("whatshap.readselect", fromlist=["readselection"]).readselection(rs, 3, preferred)
    assert shim.stats()["read_selections"] == 2
    previous.restore()   # every rebound name goes back, the heuristic factory and the read selection included
    assert (phase.Pedigree, phase.PedigreeDPTable, phase.PedMecHeuristic, phase.readselection) == (ref.Pedigree, ref.PedigreeDPTable, ref.PedMecHeuristic, reference_readselection)
    previous = shim.install(phase, ref)
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
    from whatshap_amd import ingest

    if ingest.load() is None:
        with pytest.raises(TypeError):
            phase.PedigreeDPTable(ref.ReadSet(), [1, 1, 1], ref.Pedigree(ids))  # a pedigree that was not recorded
    else:
        # with the compiled ingestion a plain reference Pedigree is read through thisptr: no recording needed (with a GPU the
        # factory returns the device table; without one it raises WHAMD_ERR_DEVICE -- never a silent CPU run)
        plain = ref.Pedigree(ids)
        plain.add_individual("father", gts)
        from whatshap_amd import _native

        if _native.device_count() > 0:
            assert phase.PedigreeDPTable(ref.ReadSet(), [1, 1, 1], plain) is not None
        else:
            with pytest.raises(_native.SolverError) as e:
                phase.PedigreeDPTable(ref.ReadSet(), [1, 1, 1], plain)
            assert e.value.status == _native.WHAMD_ERR_DEVICE


def test_shim_fallback_is_opt_in_counted_and_never_taken_for_device_errors(monkeypatch):
    """VERDICT r2 #10 / ADVICE r2: the CPU fallback of the integration shim is OPT-IN (`allow_cpu_fallback=True`), taken
    only for the device path's INPUT limits (UNSUPPORTED / OVERFLOW), counted (`shim.stats()`), and NEVER taken for
    WHAMD_ERR_DEVICE (no GPU, a HIP fault: a broken installation must surface).  Algorithmic errors are re-raised."""
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
    # default: no fallback at all
    phase = types.SimpleNamespace(Pedigree=core.Pedigree, PedigreeDPTable=FakeReferenceTable)
    previous = shim.install(phase, None)
    assert previous == (core.Pedigree, FakeReferenceTable)
    for status in (_native.WHAMD_ERR_UNSUPPORTED, _native.WHAMD_ERR_OVERFLOW, _native.WHAMD_ERR_DEVICE):
        monkeypatch.setattr(core, "PedigreeDPTable", refusing(status, "beyond the device path"))
        with pytest.raises(_native.SolverError):
            phase.PedigreeDPTable(rs, [1, 2], ped, True, [10, 20])
    assert not calls
    # opt-in: input limits fall back and are counted; device errors still raise
    phase = types.SimpleNamespace(Pedigree=core.Pedigree, PedigreeDPTable=FakeReferenceTable)
    shim.install(phase, None, allow_cpu_fallback=True)
    shim.reset_stats()
    for status in (_native.WHAMD_ERR_UNSUPPORTED, _native.WHAMD_ERR_OVERFLOW):
        monkeypatch.setattr(core, "PedigreeDPTable", refusing(status, "beyond the device path"))
        table = phase.PedigreeDPTable(rs, [1, 2], ped, True, [10, 20])
        assert isinstance(table, FakeReferenceTable) and calls[-1] == (rs, (1, 2), ped, True, [10, 20])
    st = shim.stats()
    assert st["cpu_fallbacks"] == 2 and st["device_tables"] == 0 and len(st["fallback_reasons"]) == 2
    monkeypatch.setattr(core, "PedigreeDPTable", refusing(_native.WHAMD_ERR_DEVICE, "no HIP device visible"))
    with pytest.raises(_native.SolverError, match="no HIP device"):
        phase.PedigreeDPTable(rs, [1, 2], ped)
    monkeypatch.setattr(core, "PedigreeDPTable", refusing(_native.WHAMD_ERR_MENDELIAN_CONFLICT, "Error: Mendelian conflict"))
    with pytest.raises(RuntimeError, match="Mendelian conflict"):
        phase.PedigreeDPTable(rs, [1, 2], ped)
    assert shim.stats()["cpu_fallbacks"] == 2


def test_shim_without_a_device_raises_instead_of_running_on_the_cpu():
    """The real library on a box without a GPU (this container; skipped where one is visible): shim.install with the
    fallback ALLOWED still raises WHAMD_ERR_DEVICE -- a mis-configured installation never runs 100 % on the CPU reference."""
    import types

    from whatshap_amd import _native, core, shim
    from whatshap_amd.core import Read, ReadSet

    if _native.device_count() > 0:
        pytest.skip("a HIP device is visible")

    class MustNotBeUsed:
        def __init__(self, *args, **kwargs):
            raise AssertionError("fell back to the CPU class on a device error")

    phase = types.SimpleNamespace(Pedigree=core.Pedigree, PedigreeDPTable=MustNotBeUsed)
    shim.install(phase, None, allow_cpu_fallback=True)
    ids = core.NumericSampleIds()
    ped = core.Pedigree(ids)
    ped.add_individual("a", [core.Genotype([0, 1]), core.Genotype([0, 1])])
    rs = ReadSet()
    r = Read("r", 50, 0, ids["a"])
    r.add_variant(10, 0, 5)
    r.add_variant(20, 1, 5)
    rs.add(r)
    rs.sort()
    with pytest.raises(_native.SolverError) as e:
        phase.PedigreeDPTable(rs, [1, 1], ped)
    assert e.value.status == _native.WHAMD_ERR_DEVICE


def _compiled_ingestion():
    from whatshap_amd import ingest
    from whatshap_amd.ingest import build as ingest_build

    reference_core()  # whatshap.core loaded (RTLD_GLOBAL) -- the C++ symbols the extension resolves against
    if not ingest_build.available():
        pytest.skip("whatshap_amd/ingest not built (reference tree absent when build() ran)")
    mod = ingest.load()
    assert mod is not None, "whamd_ingest was built but does not import"
    return mod


def _same_arrays(a, b):
    for name in ("read_ptr", "var_position", "var_allele", "var_quality", "read_sample_id", "individual_id", "triple_ids",
                 "genotype", "recombcost"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert a.n_variants == b.n_variants
    if a.genotype_likelihoods is None:
        assert b.genotype_likelihoods is None
    else:
        assert np.array_equal(a.genotype_likelihoods, b.genotype_likelihoods, equal_nan=True)


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_compiled_ingestion_equals_python_walk(case):
    """whatshap_amd.ingest (C++ against the reference's headers, thisptr walk as in readselect.pyx:14-15,244) produces the
    arrays the Python-level walk of the same reference objects produces -- for the ReadSet and for the Pedigree, which is
    opaque from Python (here the Python side needs the recording subclass, the compiled side does not)."""
    from whatshap_amd.core import problem_from_reference_objects

    mod = _compiled_ingestion()
    ref = reference_core()
    rs, ped = to_reference(case, ref)
    want = problem_from_objects(rs, case.recombcost, ped.amd, case.distrust_genotypes, case.positions)
    got = problem_from_reference_objects(mod, rs, case.recombcost, ped, case.distrust_genotypes, case.positions)
    _same_arrays(want, got)


def test_compiled_ingestion_on_a_plain_reference_pedigree_and_large_readset():
    """No recording subclass at all: a plain whatshap.core.Pedigree (trio with likelihoods) and a 3 000-variant ReadSet."""
    from refobjects import problem_to_reference
    from whatshap_amd.core import problem_from_reference_objects
    from whatshap_amd.synthetic import synthetic_block

    mod = _compiled_ingestion()
    ref = reference_core()
    problem = synthetic_block(3000, 9, seed=5, trio=True, distrust_genotypes=True)
    rs, recording = problem_to_reference(problem, ref)
    ids = ref.NumericSampleIds()
    for numeric in range(3):
        assert ids[str(numeric)] == numeric
    plain = ref.Pedigree(ids)
    n_var = problem.n_variants
    gl = problem.genotype_likelihoods.reshape(3, n_var, 3)
    for i in range(3):
        plain.add_individual(str(i), [ref.Genotype([0, 1])] * n_var,
                             [ref.PhredGenotypeLikelihoods([float(x) for x in gl[i, v]]) for v in range(n_var)])
    plain.add_relationship("0", "1", "2")
    assert type(plain) is ref.Pedigree
    got = problem_from_reference_objects(mod, rs, problem.recombcost.tolist(), plain, True, problem.positions.tolist())
    want = problem_from_objects(rs, problem.recombcost.tolist(), recording.amd, True, problem.positions.tolist())
    _same_arrays(want, got)
    assert int(oracle.OracleTable(got).optimal_score()) == int(oracle.OracleTable(problem).optimal_score())


def test_ingest_refuses_the_objects_of_another_whatshap_release(monkeypatch):
    """whamd_ingest reads C++ objects through thisptr with the class layouts of the release it was built against
    (recorded by build.py); with a different installed release load() declines (public-API walk instead), version stubs of
    builds without metadata ('0+oracle') are not compared."""
    import sys
    import types
    import warnings

    from whatshap_amd import ingest

    assert ingest._same_release("2.8", "2.8.1") and not ingest._same_release("2.8", "2.9.dev1+g0")
    assert ingest._is_release("2.8") and not ingest._is_release("0+oracle") and not ingest._is_release(None)
    _compiled_ingestion()   # built and loadable here
    monkeypatch.setattr(ingest, "_module", None)
    monkeypatch.setattr(ingest, "_tried", False)
    monkeypatch.setitem(sys.modules, "whatshap", types.SimpleNamespace(__version__="1.7"))
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        assert ingest.load() is None
    assert any("built against WhatsHap" in str(w.message) for w in caught)


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_compiled_emit_of_superreads_equals_the_reference_class(case):
    """whamd_ingest.emit_superreads (C++ `new Read` / `addVariant` / `new ReadSet`, adopted as whatshap/core.pyx:388-400 does) on
    the arrays of the C-ABI getter: names, sample ids, source ids, mapqs and every (position, allele, quality) of the objects equal
    what whatshap.core.PedigreeDPTable.get_super_reads returns -- here with the oracle's arrays (CPU); test_gpu_parity.py does the
    same with the device's through shim.table_factory."""
    from whatshap_amd import ingest

    ref = reference_core()
    compiled = ingest.load()
    if compiled is None or not hasattr(compiled, "emit_superreads"):
        pytest.skip("compiled ingestion not built")
    rs, ped = to_reference(case, ref)
    want = table_outputs(ref.PedigreeDPTable(rs, case.recombcost, ped, case.distrust_genotypes, case.positions))
    problem = problem_from_objects(rs, case.recombcost, ped.amd, case.distrust_genotypes, case.positions)
    table = oracle.OracleTable(problem)
    a0, a1, q, tv, sid = table.super_reads()
    sets = compiled.emit_superreads(table.positions(), a0, a1, q, sid)
    assert all(type(s) is ref.ReadSet for s in sets) and len(sets) == len(sid)
    got = []
    for readset in sets:
        assert len(readset) == 2
        for read in readset:
            got.append((read.name, read.sample_id, read.source_id, tuple(read.mapqs), [(v.position, v.allele, v.quality) for v in read]))
    assert got == want["superreads"]


def test_compiled_emit_edge_cases_and_heuristic_names():
    from whatshap_amd import ingest

    ref = reference_core()
    compiled = ingest.load()
    if compiled is None or not hasattr(compiled, "emit_superreads"):
        pytest.skip("compiled ingestion not built")
    empty = compiled.emit_superreads(np.zeros(0, np.uint32), np.zeros((2, 0), np.uint8), np.zeros((2, 0), np.uint8), np.zeros((2, 0), np.uint32), [7, 9])
    assert [[(r.name, r.sample_id, len(r)) for r in s] for s in empty] == [[("superread_0_0", 7, 0), ("superread_1_0", 7, 0)],
                                                                          [("superread_0_1", 9, 0), ("superread_1_1", 9, 0)]]
    assert compiled.emit_superreads(np.zeros(0, np.uint32), np.zeros((0, 0), np.uint8), np.zeros((0, 0), np.uint8), np.zeros((0, 0), np.uint32), []) == []
    plain = compiled.emit_superreads(np.array([5, 9], np.uint32), np.array([[0, 3]], np.uint8), np.array([[1, 3]], np.uint8),
                                     np.array([[30, 30]], np.uint32), [4], numbered=False)   # PedMecHeuristic::getSuperReads' names
    assert [(r.name, [(v.position, v.allele, v.quality) for v in r]) for r in plain[0]] == [("superread_0", [(5, 0, 30), (9, 3, 30)]),
                                                                                           ("superread_1", [(5, 1, 30), (9, 3, 30)])]
    with pytest.raises(ValueError):
        compiled.emit_superreads(np.zeros(3, np.uint32), np.zeros((1, 2), np.uint8), np.zeros((1, 2), np.uint8), np.zeros((1, 2), np.uint32), [0])
    # the point of compiling it: 200 000 columns in milliseconds, not the 254 ms of a Python loop per individual
    import time

    n = 200_000
    pos = np.arange(n, dtype=np.uint32) * 1000
    a = np.zeros((1, n), np.uint8)
    t0 = time.perf_counter()
    big = compiled.emit_superreads(pos, a, 1 - a, np.full((1, n), 17, np.uint32), [0])
    dt = time.perf_counter() - t0
    assert len(big[0][1]) == n and big[0][1][n - 1].allele == 1
    assert dt < 0.1, f"compiled emit took {dt * 1e3:.1f} ms for one individual x 200 000 columns"
