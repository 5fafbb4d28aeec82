"""GPU (-m gpu): GenotypeDPTable on the device against the REAL reference class
(whatshap.core.GenotypeDPTable, built into oracle/_ref/cy) and against the CPU restatement.  The reference computes in
long double, the device in f64: agreement to rtol 1e-9 (observed ~1e-13)."""
import numpy as np
import pytest

from genotype_cases import random_case, reference_likelihoods
from oracle import genotype_oracle
from refobjects import reference_core
from test_genotype_oracle import REFERENCE_VECTORS, _matrix_problem
from whatshap_amd import _native, core
from whatshap_amd.genotype import GenotypeDPTable

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-13


def device_likelihoods(problem, window=0):
    n_columns = problem.positions.size if problem.positions is not None else np.unique(problem.var_position).size
    gl, stats = _native.genotype_likelihoods(problem, int(n_columns), window=window)
    return gl, stats


@pytest.mark.parametrize("lines,want", REFERENCE_VECTORS, ids=["exact1", "exact2", "exact3"])
def test_known_answers_of_the_reference_tests(lines, want):
    gl, _ = device_likelihoods(_matrix_problem(lines))
    assert np.allclose(gl[0], want, rtol=1e-9)


@pytest.mark.parametrize("mode,n_cases", [("single", 40), ("trio", 25), ("quartet", 8)])
def test_small_random_cases_vs_restatement_and_reference(mode, n_cases):
    ref = reference_core()
    for seed in range(n_cases):
        p = random_case(7000 + 100 * len(mode) + seed, n_variants=8, n_reads=10 if mode == "single" else 8, max_len=4, mode=mode,
                        max_coverage=6 if mode == "single" else 4, uniform_prior=seed % 3 == 0)
        want = reference_likelihoods(p, ref)
        oracle_gl = np.asarray(genotype_oracle.genotype_likelihoods(p), dtype=np.float64)
        for window in (0, 1, 3):   # 0: the run-fused path where every column fits a run; a forced window: the per-column kernels
            got, _ = device_likelihoods(p, window)
            assert np.allclose(got, want, rtol=RTOL, atol=ATOL), (mode, seed, window, np.abs(got - want).max())
            assert np.allclose(got, oracle_gl, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("kw", [
    dict(mode="single", n_variants=300, n_reads=900, max_len=12, max_coverage=12),
    dict(mode="single", n_variants=120, n_reads=500, max_len=9, max_coverage=15, phred=(0, 60)),
    dict(mode="trio", n_variants=200, n_reads=500, max_len=8, max_coverage=9),
    dict(mode="quartet", n_variants=60, n_reads=150, max_len=6, max_coverage=7),
], ids=str)
def test_larger_cases_vs_the_reference_class(kw):
    ref = reference_core()
    p = random_case(99, **kw)
    want = reference_likelihoods(p, ref)
    for window in (0, 7):
        got, stats = device_likelihoods(p, window)
        assert np.allclose(got, want, rtol=RTOL, atol=ATOL), (window, np.abs(got - want).max())
        assert np.allclose(got.sum(axis=2), 1.0)
        # both device paths were exercised: the run-fused one whenever every column fits a run (the planner says), never under a forced window
        plan = _native.plan_summary(p, "genotype_slots")
        eligible = plan["invariants_ok"] == 1 and plan["n_resident_columns"] == plan["n_columns"]
        assert (stats["slot_runs"] > 0) == (window == 0 and eligible), (stats, plan)
    assert stats["n_columns"] == kw["n_variants"]


def _synthetic_genotyping_problem(n_variants, coverage, seed, **kw):
    """A synthetic phasing block (whatshap_amd.synthetic) with random genotype priors: long tables at a steady coverage."""
    from whatshap_amd.synthetic import synthetic_block

    b = synthetic_block(n_variants, coverage, seed=seed, **kw)
    rng = np.random.default_rng(seed)
    gl = rng.random((b.n_individuals, b.n_variants, 3)) + 0.05
    gl /= gl.sum(axis=2, keepdims=True)
    return _native.ProblemArrays(b.read_ptr, b.var_position, b.var_allele, b.var_quality, b.read_sample_id, b.individual_id, b.triple_ids,
                                 b.genotype.reshape(b.n_individuals, -1), gl, b.recombcost, b.positions, False, n_variants=b.n_variants)


@pytest.mark.parametrize("kw", [dict(n_variants=6000, coverage=11, seed=3), dict(n_variants=1500, coverage=9, seed=4, trio=True)], ids=str)
def test_no_drift_over_thousands_of_columns(kw):
    """f64 on the device against the long-double reference class over thousands of normalised columns (the runs
    rescale once per launch, the reference once per column): still rtol 1e-9 at the far end of the table, on both device paths."""
    ref = reference_core()
    p = _synthetic_genotyping_problem(**kw)
    want = reference_likelihoods(p, ref)
    got, stats = device_likelihoods(p)
    assert stats["slot_runs"] > 100, stats
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL), np.abs(got - want).max()
    tail = slice(-50, None)
    assert np.allclose(got[:, tail], want[:, tail], rtol=RTOL, atol=ATOL)
    old, old_stats = device_likelihoods(p, window=256)
    assert old_stats["slot_runs"] == 0
    assert np.allclose(old, want, rtol=RTOL, atol=ATOL), np.abs(old - want).max()


@pytest.mark.parametrize("kw,window_bytes", [(dict(n_variants=5000, coverage=11, seed=5), 24 << 20), (dict(n_variants=1200, coverage=9, seed=6, trio=True), 4 << 20),
                                             (dict(n_variants=1200, coverage=9, seed=6, trio=True), 3 << 19)], ids=str)
def test_windowed_run_path_is_identical_to_the_unwindowed_one(kw, window_bytes, monkeypatch):
    """Column stores larger than their budget (WHAMD_GENO_WINDOW_BYTES stands in for HBM): the runs are cut into windows, the forward
    columns of every window but the newest are recomputed from the kept exchange column (src/genotypedptable.cpp:116-157,159-195,324 keeps
    sqrt(n) columns for the same reason).  Same operations in the same order: the likelihoods are the SAME doubles, and they equal the
    reference class's to rtol 1e-9."""
    p = _synthetic_genotyping_problem(**kw)
    whole, stats = device_likelihoods(p)
    assert stats["slot_runs"] > 50 and stats["window"] == p.n_variants
    monkeypatch.setenv("WHAMD_GENO_WINDOW_BYTES", str(window_bytes))
    windowed, wstats = device_likelihoods(p)
    assert wstats["slot_runs"] == stats["slot_runs"] and wstats["window"] < p.n_variants // 3, wstats
    assert np.array_equal(windowed, whole), np.abs(windowed - whole).max()
    want = reference_likelihoods(p, reference_core())
    assert np.allclose(windowed, want, rtol=RTOL, atol=ATOL), np.abs(windowed - want).max()


def test_many_reads_starting_and_ending_in_one_column():
    """Columns in which more reads start / end than a thread loops over: the split (atomic) accumulation path."""
    ref = reference_core()
    rng = np.random.default_rng(5)
    read_ptr, pos, alle, qual = [0], [], [], []
    for r in range(9):                      # nine reads over the same three variants, then three over the next two
        for v in (10, 20, 30):
            pos.append(v); alle.append(int(rng.integers(0, 2))); qual.append(int(rng.integers(5, 30)))
        read_ptr.append(len(pos))
    for r in range(3):
        for v in (40, 50):
            pos.append(v); alle.append(int(rng.integers(0, 2))); qual.append(int(rng.integers(5, 30)))
        read_ptr.append(len(pos))
    n_var = 5
    gl = rng.random((1, n_var, 3)) + 0.1
    gl /= gl.sum(axis=2, keepdims=True)
    p = _native.ProblemArrays(np.asarray(read_ptr, dtype=np.uint64), np.asarray(pos, dtype=np.int32), np.asarray(alle, dtype=np.uint8),
                              np.asarray(qual, dtype=np.uint32), np.zeros(12, dtype=np.int32), np.asarray([0], dtype=np.uint32),
                              np.zeros(0, dtype=np.uint32), np.ones((1, n_var), dtype=np.uint8), gl, np.full(n_var, 7, dtype=np.uint32), None, False,
                              n_variants=n_var)
    want = reference_likelihoods(p, ref)
    got, _ = device_likelihoods(p)
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL), np.abs(got - want).max()


def test_python_class_mirrors_the_reference_constructor():
    """whatshap_amd.genotype.GenotypeDPTable with the mirror ReadSet / Pedigree (tests/test_genotyping.py:66-95)."""
    from helpers import string_to_readset

    readset = string_to_readset("11\n 01", None, scale_quality=10)
    positions = readset.get_positions()
    ids = core.NumericSampleIds()
    pedigree = core.Pedigree(ids)
    pedigree.add_individual("individual0", [core.Genotype([0, 1])] * len(positions),
                            [core.PhredGenotypeLikelihoods([1 / 3, 1 / 3, 1 / 3])] * len(positions))
    table = GenotypeDPTable(ids, readset, [1] * len(positions), pedigree)
    want = REFERENCE_VECTORS[0][1]
    for c in range(len(positions)):
        got = table.get_genotype_likelihoods("individual0", c)
        assert np.allclose(got.as_vector(), want[c], rtol=1e-9)
    assert table.get_stats()["n_columns"] == 3


def test_empty_readset_and_missing_priors():
    ids = core.NumericSampleIds()
    pedigree = core.Pedigree(ids)
    pedigree.add_individual("individual0", [core.Genotype([0, 1])] * 2, [None, None])
    GenotypeDPTable(ids, core.ReadSet(), [1, 1], pedigree)   # tests/test_genotyping.py:54-62: nothing to do, no error
    from helpers import string_to_readset

    readset = string_to_readset("11\n01", None, scale_quality=10)
    with pytest.raises(_native.SolverError, match="priors"):
        GenotypeDPTable(ids, readset, [1, 1], pedigree)


def test_shim_runs_the_reference_objects_through_the_device():
    """whatshap.cli.genotype's call (cli/genotype.py:357-368) with WhatsHap's OWN ReadSet / Pedigree objects, rebound through
    shim.install_genotype: same likelihoods as the reference class, returned as the reference's PhredGenotypeLikelihoods."""
    import types

    from whatshap_amd import shim

    ref = reference_core()
    module = types.SimpleNamespace(Pedigree=ref.Pedigree, GenotypeDPTable=ref.GenotypeDPTable)
    previous = shim.install_genotype(module, ref)
    assert previous[1] is ref.GenotypeDPTable
    p = random_case(4242, n_variants=40, n_reads=120, max_len=7, mode="trio", max_coverage=9)
    rs = ref.ReadSet()
    for r in range(p.n_reads):
        read = ref.Read(f"read{r}", 60, 0, int(p.read_sample_id[r]))
        for i in range(int(p.read_ptr[r]), int(p.read_ptr[r + 1])):
            read.add_variant(int(p.var_position[i]), int(p.var_allele[i]), int(p.var_quality[i]))
        rs.add(read)
    ids = ref.NumericSampleIds()
    for name in ("0", "1", "2"):
        ids[name]
    gl = p.genotype_likelihoods.reshape(3, p.n_variants, 3)
    peds = []
    for cls in (module.Pedigree, ref.Pedigree):
        ped = cls(ids)
        for i in range(3):
            ped.add_individual(str(i), [ref.Genotype([0, 1])] * p.n_variants,
                               [ref.PhredGenotypeLikelihoods([float(x) for x in gl[i, v]]) for v in range(p.n_variants)])
        ped.add_relationship("0", "1", "2")
        peds.append(ped)
    positions = [int(x) for x in p.positions]
    recomb = [int(x) for x in p.recombcost]
    ours = module.GenotypeDPTable(ids, rs, recomb, peds[0], positions)
    theirs = ref.GenotypeDPTable(ids, rs, recomb, peds[1], positions)
    for name in ("0", "1", "2"):
        for c in range(len(positions)):
            a, b = ours.get_genotype_likelihoods(name, c), theirs.get_genotype_likelihoods(name, c)
            assert type(a) is type(b)
            assert np.allclose(list(a), list(b), rtol=RTOL, atol=ATOL)


def test_committed_golden_vectors():
    from test_genotype_oracle import golden_cases

    for name, problem, want in golden_cases():
        got, _ = device_likelihoods(problem)
        assert np.allclose(got, want, rtol=RTOL, atol=ATOL), (name, np.abs(got - want).max())
