"""CPU: the pedigree slot plan (whatshap_amd/csrc/slot_plan.cpp on a table with one or two trios) checked against the
oracle without a GPU.  `whamd_debug_emulate_pedslot_plan` executes the plan the way kernels_pedslots.h does -- cost forms
split into per-workgroup / per-wave / per-lane tables, one (cell, transmission value) per lane, the butterfly min-plus
step over the previous transmission value, pair decisions of the ending reads, one record byte per lane and column, the
column-by-column walk -- and cost, index path and transmission path must equal the oracle's."""
import random

import numpy as np
import pytest

import oracle
from whatshap_amd import _native
from whatshap_amd.synthetic import random_small_instance, synthetic_block


def agrees(problem, **kw):
    o = oracle.OracleTable(problem)
    want_idx, want_trans = o.index_path()
    idx, trans, score, run_columns = _native.emulate_pedslot_plan(problem, o.n_columns, **kw)
    ok = score == o.optimal_score() and bool((idx == want_idx).all()) and bool((trans == want_trans).all())
    return ok, run_columns


@pytest.mark.parametrize("mode", ["trio", "quartet"])
def test_random_tie_heavy_instances(mode):
    rng = random.Random(5 if mode == "trio" else 6)
    eligible = with_runs = 0
    for i in range(400):
        p = random_small_instance(rng, mode=mode, allow_conflict=False)
        try:
            ok, run_columns = agrees(p, slot_l=0 if i % 2 else (4 if mode == "trio" else 2))
        except _native.SolverError as e:
            assert e.status == _native.WHAMD_ERR_UNSUPPORTED   # genotypes not trusted: more than 4 forms per value
            continue
        eligible += 1
        with_runs += run_columns > 0
        assert ok, i
    assert eligible > 80 and with_runs > 80, (eligible, with_runs)


@pytest.mark.parametrize("seed", range(4))
def test_synthetic_trio_blocks(seed):
    for kw in (dict(coverage=7 + seed % 3), dict(coverage=8, step=1), dict(coverage=7, mixed_genotypes=True), dict(coverage=9, step=3)):
        base = synthetic_block(n_variants=90, seed=seed, trio=True, **kw)
        ties = _native.ProblemArrays(base.read_ptr, base.var_position, base.var_allele, (1 + (base.var_quality % 2)).astype(np.uint32),
                                     base.read_sample_id, base.individual_id, base.triple_ids, base.genotype.reshape(base.individual_id.size, -1), base.genotype_likelihoods,
                                     (base.recombcost % 3).astype(np.uint32), base.positions, base.distrust_genotypes, n_variants=base.n_variants)
        for p in (base, ties):
            for slot_l in (0, 5, 6):
                ok, run_columns = agrees(p, slot_l=slot_l)
                assert ok, (seed, kw, slot_l)
                assert run_columns > 0.7 * p.n_variants, (kw, run_columns)


@pytest.mark.parametrize("seed", range(3))
def test_synthetic_quartet_blocks(seed):
    for kw in (dict(coverage=6 + seed), dict(coverage=7, mixed_genotypes=True)):
        p = synthetic_block(n_variants=70, seed=10 + seed, quartet=True, **kw)
        for slot_l in (0, 3):
            ok, run_columns = agrees(p, slot_l=slot_l)
            assert ok, (seed, kw, slot_l)
            assert run_columns > 0.6 * p.n_variants, (kw, run_columns)


@pytest.mark.parametrize("seed", range(2))
def test_two_unrelated_trios_need_four_forms_per_value(seed):
    """Six individuals, four founders: 2^4 allele assignments, two children's constraints leave 4 -- the NF = 4 tables."""
    for kw in (dict(coverage=6), dict(coverage=8, mixed_genotypes=True)):
        p = synthetic_block(n_variants=60, seed=20 + seed, two_trios=True, **kw)
        ok, run_columns = agrees(p)
        assert ok, (seed, kw)
        assert run_columns > 0.6 * p.n_variants, (kw, run_columns)


def test_not_for_a_single_individual():
    single = synthetic_block(n_variants=40, coverage=6, seed=1)
    with pytest.raises(_native.SolverError) as e:
        _native.emulate_pedslot_plan(single, 40)
    assert e.value.status == _native.WHAMD_ERR_UNSUPPORTED


@pytest.mark.parametrize("kw", [dict(n_variants=60, coverage=6, seed=1, trio=True, distrust_genotypes=True),
                                dict(n_variants=300, coverage=8, seed=2, trio=True, distrust_genotypes=True),
                                dict(n_variants=200, coverage=9, seed=3, trio=True, distrust_genotypes=True, mixed_genotypes=True)], ids=str)
@pytest.mark.parametrize("factorised", [True, False], ids=["factorised", "sixteen-forms"])
def test_untrusted_genotypes_of_a_trio_on_sixteen_forms(kw, factorised, monkeypatch):
    """Genotypes not trusted: up to 15 distinct cost forms per transmission value (16 allele assignments, src/pedigreecolumncostcomputer.cpp:14-50).
    Default: the FACTORISED line (slots.h PSLOT_FACT: the untransmitted alleles minimised out first, three sums + twelve constants, checked
    against the generic term list when the problem is built); WHAMD_NO_PED_FACT: runs with NF = 16.  Plan, tables, records and walk emulated
    on the CPU equal the oracle in both."""
    if not factorised:
        monkeypatch.setenv("WHAMD_NO_PED_FACT", "1")
    p = synthetic_block(**kw)
    summary = _native.plan_summary(p, "slots")
    assert summary["invariants_ok"] == 1
    assert (summary["n_fact_runs"] == summary["n_runs"]) if factorised else (summary["n_fact_runs"] == 0), summary
    ok, run_columns = agrees(p)
    assert ok, kw
    assert run_columns > 0.8 * p.n_variants, (kw, run_columns)


def test_factorised_lines_with_any_role_order_and_uneven_likelihoods():
    """The child may be any of the three individuals: the roles (which founder transmits to which haplotype of the child, on which of the
    founder's haplotypes the transmitted allele sits) are read off the haplotype-to-partition map, for genotype likelihoods that differ
    per individual, genotype and column."""
    rng = np.random.default_rng(5)
    base = synthetic_block(n_variants=120, coverage=8, seed=11, trio=True, distrust_genotypes=True)
    ids = [int(v) for v in base.individual_id]
    for roles in ([0, 1, 2], [2, 0, 1], [1, 2, 0], [0, 2, 1]):   # (mother, father, child) as positions in individual_id
        gl = rng.integers(0, 60, size=base.genotype_likelihoods.shape).astype(np.float64)
        p = _native.ProblemArrays(base.read_ptr, base.var_position, base.var_allele, base.var_quality, base.read_sample_id, base.individual_id,
                                  np.array([ids[r] for r in roles], dtype=np.uint32), base.genotype, gl, base.recombcost, base.positions, True,
                                  n_variants=base.n_variants)
        summary = _native.plan_summary(p, "slots")
        assert summary["invariants_ok"] == 1 and summary["n_fact_runs"] == summary["n_runs"] > 0, (roles, summary)
        ok, _ = agrees(p)
        assert ok, roles


@pytest.mark.parametrize("kw", [dict(n_variants=60, coverage=6, seed=1, quartet=True, distrust_genotypes=True),
                                dict(n_variants=200, coverage=7, seed=2, quartet=True, distrust_genotypes=True),
                                dict(n_variants=150, coverage=8, seed=3, quartet=True, distrust_genotypes=True, mixed_genotypes=True)], ids=str)
def test_untrusted_genotypes_of_a_quartet_on_factorised_lines(kw):
    """Two children of the same two founders, genotypes not trusted (T = 16, sixteen allele assignments per value,
    src/pedigreecolumncostcomputer.cpp:14-50): sixteen forms per (column, value, lane) would be four columns per run, so these tables ran on the
    per-column kernels.  In haplotype space the line does not depend on the transmission value (slots.h PSLOT_FACT4: four signed sums and sixteen
    constants per COLUMN; the value only wires the children to the founders' haplotypes) and the minimum is taken by elimination -- every run of
    the plan is such a run, and plan, tables, records and walk emulated on the CPU equal the oracle."""
    p = synthetic_block(**kw)
    summary = _native.plan_summary(p, "slots")
    assert summary["invariants_ok"] == 1 and summary["n_fact_runs"] == summary["n_runs"] > 0, summary
    ok, run_columns = agrees(p)
    assert ok, kw
    assert run_columns > 0.8 * p.n_variants, (kw, run_columns)


def test_a_quartets_factorised_lines_with_any_role_order_and_uneven_likelihoods():
    """Which individuals are the founders, which the children, and the order of the two trios are read off the haplotype-to-partition maps; the
    genotype likelihoods differ per individual, genotype and column.  A pedigree the wiring does not describe (the founders swap their roles
    between the trios) keeps the generic term list -- and is still solved exactly by the plan's per-column steps."""
    rng = np.random.default_rng(9)
    base = synthetic_block(n_variants=90, coverage=7, seed=21, quartet=True, distrust_genotypes=True)
    ids = [int(v) for v in base.individual_id]
    for triples in ([0, 1, 2, 0, 1, 3], [0, 1, 3, 0, 1, 2], [2, 3, 0, 2, 3, 1], [3, 1, 0, 3, 1, 2], [1, 2, 3, 1, 2, 0]):   # (mother, father, child) x 2 as positions in individual_id
        gl = rng.integers(0, 60, size=base.genotype_likelihoods.shape).astype(np.float64)
        p = _native.ProblemArrays(base.read_ptr, base.var_position, base.var_allele, base.var_quality, base.read_sample_id, base.individual_id,
                                  np.array([ids[r] for r in triples], dtype=np.uint32), base.genotype, gl, base.recombcost, base.positions, True,
                                  n_variants=base.n_variants)
        summary = _native.plan_summary(p, "slots")
        assert summary["invariants_ok"] == 1 and summary["n_fact_runs"] == summary["n_runs"] > 0, (triples, summary)
        ok, _ = agrees(p)
        assert ok, triples


def _trio_reads_problem(reads, n_variants, seed):
    """A trio problem from explicit reads [(first variant, last variant)], samples round-robin, alleles / qualities seeded."""
    rng = np.random.default_rng(seed)
    reads = sorted(reads)
    ptr, pos, allele, qual = [0], [], [], []
    for first, last in reads:
        for v in range(first, last + 1):
            pos.append(100 * (v + 1)); allele.append(int(rng.integers(0, 2))); qual.append(int(rng.integers(1, 4)))
        ptr.append(len(pos))
    samples = np.arange(len(reads), dtype=np.int32) % 3
    return _native.ProblemArrays(np.array(ptr, dtype=np.uint64), np.array(pos, dtype=np.int32), np.array(allele, dtype=np.uint8), np.array(qual, dtype=np.uint32),
                                 samples, np.array([0, 1, 2], dtype=np.uint32), np.array([0, 1, 2], dtype=np.uint32),
                                 np.ones((3, n_variants), dtype=np.uint8), None, np.array([0] + [int(rng.integers(1, 6)) for _ in range(n_variants - 1)], dtype=np.uint32),
                                 np.array([100 * (v + 1) for v in range(n_variants)], dtype=np.uint32), False, n_variants=n_variants)


@pytest.mark.parametrize("seed", range(4))
def test_four_reads_ending_in_one_column_of_a_pedigree_run(seed):
    """PSLOT_MAXEND = 4: the fourth decision bit of a lane's record byte, the fourth ending read's slot next to the count in the
    backtrace column; the run continues through such a column."""
    rng = np.random.default_rng(60 + seed)
    n = 36
    reads = []
    for stop in (8, 15, 23, 30):
        for q in range(4):
            reads.append((int(stop - 2 - rng.integers(0, 4)), stop))
    for first in range(0, n - 6, 4):
        reads.append((first, min(n - 1, first + int(rng.integers(5, 10)))))
    p = _trio_reads_problem(reads, n, seed)
    for slot_l in (0, 5, 6):
        ok, run_columns = agrees(p, slot_l=slot_l)
        assert ok, (seed, slot_l)
        assert run_columns >= n - 5, run_columns


def test_lazy_generic_terms_equal_the_eager_ones():
    """Round 6 (csrc/problem.cpp `lazy_fact_terms`, fill_lazy_terms): whamd_dptable_create builds the generic term lists of a trio / quartet with untrusted genotypes
    only for a sample of columns (where the factorised line is checked) and, after planning, for the columns outside runs.  Whatever columns are asked for, in however
    many calls: the lists equal the eagerly built ones term by term, the factorised line is the same, the value bound is not smaller; tables without a factorised line
    (trusted genotypes, one individual) do not take the route at all."""
    import numpy as np

    from whatshap_amd import _native
    from whatshap_amd.synthetic import synthetic_block

    for kw in (dict(trio=True, distrust_genotypes=True), dict(quartet=True, distrust_genotypes=True)):
        p = synthetic_block(700, 7, seed=11, **kw)
        n = p.n_variants
        everything = _native.debug_lazy_terms_check(p)
        assert everything["lazy"] and everything["differences"] == 0
        assert everything["built_before"] < n // 2 and everything["built_after"] == n      # the sample (first 256 + every 64th), then all
        rng = np.random.default_rng(3)
        need = (rng.random(n) < 0.1).astype(np.uint8)
        need[[0, n - 1, 300, 301]] = 1
        some = _native.debug_lazy_terms_check(p, need=need, rounds=3)                       # three fills, each merging into what is there
        assert some["lazy"] and some["differences"] == 0
        assert some["built_before"] <= some["built_after"] < n
        none = _native.debug_lazy_terms_check(p, need=np.zeros(n, np.uint8))
        assert none["differences"] == 0 and none["built_after"] == none["built_before"]
    for kw in (dict(trio=True), dict()):
        assert not _native.debug_lazy_terms_check(synthetic_block(300, 7, seed=11, **kw))["lazy"]
