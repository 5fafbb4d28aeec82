"""GPU (-m gpu): PedMecHeuristic on the device (one persistent single-workgroup kernel, whatshap_amd/csrc/heuristic_device.hip)
against the compiled reference's PedMecHeuristic and the committed golden vectors: bipartition, transmission vector,
haplotypes and mutations, identical (the float scores are restated operation by operation, so every decision of the beam is
the reference's); coverages beyond the exact DP's 25 reads per column; the drop-in class and the shim."""
import numpy as np
import pytest

import oracle
from helpers import load_golden, problem_from_json
from heuristic_cases import random_cases, result_tuple, synthetic_cases
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block

pytestmark = pytest.mark.gpu


def device(problem, row_limit):
    return _native.pedmec_heuristic(problem, row_limit=row_limit)


def test_random_instances_vs_the_compiled_reference():
    for name, problem, row_limit in random_cases(815, 50):
        want = oracle.heuristic_tuple(oracle.ReferenceHeuristic(problem, row_limit=row_limit))
        assert result_tuple(device(problem, row_limit)) == want, name


@pytest.mark.parametrize("case", synthetic_cases(), ids=lambda c: c[0])
def test_synthetic_blocks_vs_the_compiled_reference(case):
    name, problem, row_limit = case
    want = oracle.heuristic_tuple(oracle.ReferenceHeuristic(problem, row_limit=row_limit))
    got = device(problem, row_limit)
    assert result_tuple(got) == want, name
    host = _native.pedmec_heuristic(problem, row_limit=row_limit, host_diagnostic=True)
    assert got["stats"]["max_solutions"] == host["stats"]["max_solutions"] and got["stats"]["total_solutions"] == host["stats"]["total_solutions"]


def test_golden_vectors():
    for rec in load_golden("heuristic_cases.json"):
        assert result_tuple(device(problem_from_json(rec["problem"]), rec["row_limit"])) == rec["solution"], rec["name"]


@pytest.mark.parametrize("kw,row_limit", [(dict(n_variants=3000, coverage=30, seed=11), 256), (dict(n_variants=2000, coverage=24, seed=12, trio=True), 256),
                                          (dict(n_variants=1500, coverage=18, seed=13, quartet=True, error_rate=0.08), 128),
                                          (dict(n_variants=1200, coverage=40, seed=14, error_rate=0.1), 1024),
                                          (dict(n_variants=400, coverage=70, seed=15, error_rate=0.05), 64)], ids=str)
def test_coverages_the_exact_dp_cannot_afford(kw, row_limit):
    """Thousands of columns at coverage 18 - 40 (the exact table stops at 25 reads per column): the use the solver exists for."""
    p = synthetic_block(**kw)
    want = oracle.heuristic_tuple(oracle.ReferenceHeuristic(p, row_limit=row_limit))
    got = device(p, row_limit)
    assert result_tuple(got) == want
    assert got["stats"]["max_solutions"] >= min(row_limit, 16)


def test_fewer_threads_than_solutions(monkeypatch):
    """One wavefront for a beam of hundreds of solutions (WHAMD_HEURISTIC_THREADS): every phase loops over the beam in rounds."""
    p = synthetic_block(n_variants=800, coverage=26, seed=16, trio=True)
    want = oracle.heuristic_tuple(oracle.ReferenceHeuristic(p, row_limit=128))
    monkeypatch.setenv("WHAMD_HEURISTIC_THREADS", "64")
    assert result_tuple(device(p, 128)) == want


def test_drop_in_class_and_shim():
    """whatshap_amd.heuristic.PedMecHeuristic with the mirror objects (constructor order of core.pyx:675, the four getters), and
    shim.install rebinding phase.PedMecHeuristic next to PedigreeDPTable."""
    import types

    from whatshap_amd import core, shim
    from whatshap_amd.heuristic import PedMecHeuristic

    p = synthetic_block(n_variants=120, coverage=10, seed=21, trio=True)
    ids = core.NumericSampleIds()
    ped = core.Pedigree(ids)
    for name in ("father", "mother", "child"):
        ped.add_individual(name, [core.Genotype([0, 1])] * p.n_variants)
    ped.add_relationship("father", "mother", "child")
    rs = core.ReadSet()
    for r in range(p.n_reads):
        read = core.Read(f"read{r}", 60, 0, int(p.read_sample_id[r]))
        for i in range(int(p.read_ptr[r]), int(p.read_ptr[r + 1])):
            read.add_variant(int(p.var_position[i]), int(p.var_allele[i]), int(p.var_quality[i]))
        rs.add(read)
    recomb = [int(x) for x in p.recombcost]
    solver = PedMecHeuristic(rs, recomb, ped, 64, distrust_genotypes=False, positions=[int(x) for x in p.positions], allow_mutations=True, verbosity=0)
    want = oracle.ReferenceHeuristic(p, row_limit=64)
    superreads, transmission = solver.get_super_reads()
    haps, mut = want.haplotypes()
    assert transmission == want.transmission().tolist() and solver.get_optimal_cost() == 0
    assert solver.get_optimal_partitioning() == [0 if b else 1 for b in want.bipartition().tolist()]
    assert len(superreads) == 3
    for s, readset in enumerate(superreads):
        reads = list(readset)
        assert [r.name for r in reads] == ["superread_0", "superread_1"] and all(r.sample_id == s for r in reads)
        for hap, read in enumerate(reads):
            assert [v.allele for v in read] == haps[s, hap].tolist() and all(v.quality == 30 for v in read)
            assert [v.position for v in read] == p.positions.tolist()
    assert solver.get_mutations() == [[(hap, c) for c in range(p.n_variants) for hap in (0, 1) if mut[s, hap, c]] for s in range(3)]
    phase = types.SimpleNamespace(Pedigree=core.Pedigree, PedigreeDPTable=object, PedMecHeuristic=object)
    shim.install(phase, None)
    rebound = phase.PedMecHeuristic(rs, recomb, ped, 64, distrust_genotypes=False, positions=[int(x) for x in p.positions])
    assert rebound.get_super_reads()[1] == transmission and rebound.get_optimal_partitioning() == solver.get_optimal_partitioning()


def test_32_tables_in_one_launch_equal_single_solves_and_the_reference():
    """whamd_pedmec_heuristic_enqueue_many: 32 different tables (single individuals, trios, a quartet; coverages 12 - 40; row limits
    per batch) as ONE launch with one persistent workgroup each -- every output equals the table solved alone and the compiled reference's."""
    specs = []
    for i in range(20):
        specs.append(dict(n_variants=300 + 40 * i, coverage=18 + (i % 5) * 4, seed=200 + i, error_rate=0.03 + 0.01 * (i % 4)))
    for i in range(9):
        specs.append(dict(n_variants=250 + 50 * i, coverage=12 + 2 * (i % 4), seed=230 + i, trio=True))
    specs += [dict(n_variants=300, coverage=12, seed=240, quartet=True), dict(n_variants=500, coverage=40, seed=241, error_rate=0.1),
              dict(n_variants=200, coverage=16, seed=242, trio=True, distrust_genotypes=True)]
    assert len(specs) == 32
    problems = [synthetic_block(**kw) for kw in specs]
    for row_limit in (64, 256):
        got = _native.pedmec_heuristic_many(problems, row_limit=row_limit)
        assert len(got) == 32
        for i, (p, g) in enumerate(zip(problems, got)):
            if i % 4 == 0 or row_limit == 64:
                want = oracle.heuristic_tuple(oracle.ReferenceHeuristic(p, row_limit=row_limit))
                assert result_tuple(g) == want, (row_limit, specs[i])
            if i % 8 == 1:
                assert result_tuple(g) == result_tuple(device(p, row_limit)), (row_limit, specs[i])
    # a second batch reuses the device buffers of the first (the per-process pool), with another order and size
    again = _native.pedmec_heuristic_many(problems[::-3], row_limit=64)
    first = _native.pedmec_heuristic_many(problems, row_limit=64)
    for g, f in zip(again, first[::-3]):
        assert result_tuple(g) == result_tuple(f)
    _native.release_caches()
    assert result_tuple(device(problems[0], 64)) == result_tuple(first[0])
