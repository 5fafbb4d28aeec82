"""GPU (-m gpu): quartets (two children of the same two founders, T = 16) whose genotypes are not trusted, on the factorised lines of slots.h
PSLOT_FACT4 -- `pedslot_run<4, PSLOT_FACT4>` / `pedslot_group<4, PSLOT_FACT4>` -- against the oracle (the compiled reference:
src/pedigreecolumncostcomputer.cpp:14-50,101-114 enumerates the sixteen allele assignments per cell and transmission value) and against the
per-column kernels these tables ran on before.  Bit-exact (integer path)."""
import numpy as np
import pytest

import oracle
from helpers import first_difference, table_solution
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block

pytestmark = pytest.mark.gpu


def solve(problem, path="auto"):
    t = _native.NativeTable(problem, solve=False, path=path)
    t.solve()
    out, stats = table_solution(t), t.stats()
    t.close()
    return out, stats


@pytest.mark.parametrize("kw", [dict(n_variants=300, coverage=8, seed=501, quartet=True, distrust_genotypes=True),
                                dict(n_variants=400, coverage=10, seed=502, quartet=True, distrust_genotypes=True, mixed_genotypes=True),
                                dict(n_variants=300, coverage=11, seed=503, quartet=True, distrust_genotypes=True, step=1),
                                # full width: the bench's quartet (coverage 13, seed 5) with its genotypes not trusted -- 256 workgroups x 8 waves per launch
                                dict(n_variants=50000, coverage=13, seed=5, quartet=True, distrust_genotypes=True, n_columns_limit=260)], ids=str)
def test_untrusted_quartets_on_factorised_lines_vs_oracle(kw):
    p = synthetic_block(**kw)
    summary = _native.plan_summary(p)
    assert summary["n_fact_runs"] == summary["n_runs"] > 0, summary
    want = table_solution(oracle.OracleTable(p))
    got, stats = solve(p)
    assert got == want, first_difference(want, got)
    assert stats["forward_launches"] <= p.n_variants // 3, "the table did not run on pedigree slot runs"
    if p.n_variants <= 400:
        col, _ = solve(p, "column")
        assert col == want, first_difference(want, col)


@pytest.mark.parametrize("coverage", [3, 4, 5, 6, 7, 9])
def test_a_quartets_factorised_lines_in_workgroups_of_every_size(coverage, monkeypatch):
    """Coverage 3 ... 9: workgroups of 64 ... 512 threads (T = 16: a cell is sixteen lanes; the K table of a run is staged by however many threads there
    are).  WHAMD_NO_PED_FACT: the same tables on the per-column kernels; both == the oracle."""
    p = synthetic_block(n_variants=80, coverage=coverage, seed=520 + coverage, quartet=True, distrust_genotypes=True, mixed_genotypes=coverage % 2 == 0)
    want = table_solution(oracle.OracleTable(p))
    assert _native.plan_summary(p)["n_fact_runs"] > 0
    got, _ = solve(p)
    assert got == want, first_difference(want, got)
    monkeypatch.setenv("WHAMD_NO_PED_FACT", "1")
    assert _native.plan_summary(p)["n_fact_runs"] == 0
    generic, _ = solve(p)
    assert generic == want, first_difference(want, generic)


def test_a_quartets_roles_in_any_order_with_uneven_likelihoods_and_two_valued_weights():
    """Founders and children in any position of the pedigree, the two trios in either order; genotype likelihoods that differ per individual, genotype
    and column; weights from {4, 8} and cheap recombination so that nearly every minimum is attained more than once (lowest-j rule of the min-plus
    step, src/pedigreedptable.cpp:264-300; Gray-order rule of the projection, :306-327)."""
    rng = np.random.default_rng(19)
    base = synthetic_block(n_variants=260, coverage=9, seed=531, quartet=True, distrust_genotypes=True)
    ids = [int(v) for v in base.individual_id]
    for triples in ([0, 1, 2, 0, 1, 3], [0, 1, 3, 0, 1, 2], [2, 3, 0, 2, 3, 1], [3, 1, 0, 3, 1, 2]):
        gl = rng.integers(0, 40, size=base.genotype_likelihoods.shape).astype(np.float64)
        q = rng.choice(np.array([4, 8], dtype=np.uint32), size=base.var_quality.size)
        rc = rng.choice(np.array([0, 1, 2, 12], dtype=np.uint32), size=base.recombcost.size)
        p = _native.ProblemArrays(base.read_ptr, base.var_position, base.var_allele, q, base.read_sample_id, base.individual_id,
                                  np.array([ids[r] for r in triples], dtype=np.uint32), base.genotype, gl, rc, base.positions, True, n_variants=base.n_variants)
        assert _native.plan_summary(p)["n_fact_runs"] > 0, triples
        want = table_solution(oracle.OracleTable(p))
        got, _ = solve(p)
        assert got == want, (triples, first_difference(want, got))


def test_untrusted_quartets_share_their_launches():
    """Six such tables (and an untrusted trio, a trusted quartet: their own kernel variants in the same super-steps) as one sequence of launches:
    batched == the oracle for every table."""
    cases = [synthetic_block(n_variants=n, coverage=cov, seed=540 + i, quartet=True, distrust_genotypes=True, mixed_genotypes=i % 2 == 1)
             for i, (cov, n) in enumerate([(5, 200), (7, 300), (9, 400), (10, 500), (11, 600), (12, 500)])]
    cases += [synthetic_block(n_variants=400, coverage=10, seed=550, trio=True, distrust_genotypes=True), synthetic_block(n_variants=400, coverage=10, seed=551, quartet=True)]
    assert all(_native.plan_summary(p)["n_fact_runs"] > 0 for p in cases[:7])
    tables = [_native.NativeTable(p, solve=False) for p in cases]
    _native.enqueue_many(tables)
    _native.wait_many(tables)
    assert max(t.stats()["group_tables"] for t in tables) >= 6, "the untrusted quartets did not share their launches"
    for i, (p, t) in enumerate(zip(cases, tables)):
        want, got = table_solution(oracle.OracleTable(p)), table_solution(t)
        assert got == want, (i, first_difference(want, got))
        t.close()
