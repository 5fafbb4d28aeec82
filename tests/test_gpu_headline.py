"""GPU (-m gpu): bit-exact parity where the headline number is produced (BASELINE configs[2] / configs[4]).

* coverage-20 prefix of the configs[2] ReadSet, >= 300 full-width columns, against the oracle, for every forward path
  and every setting of the complement symmetry (SURVEY.md 8d parity protocol);
* coverage 21 and 23 (the CLI's cap, whatshap/cli/phase.py:1181-1182) prefixes with >= 20 full-width columns;
* at FULL size (configs[2] and one configs[4] block): every forward path / symmetry setting produces the same cost,
  index path, partitioning and superreads (the tie rules of src/pedigreedptable.cpp:306-327 included), and the optimum
  equals the independently re-evaluated objective.
"""
import numpy as np
import pytest

import oracle
from helpers import first_difference, table_solution, wmec_cost_of_partitioning
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block

pytestmark = pytest.mark.gpu

# (path, symmetry) combinations of the single-individual forward pass; "auto" is what bench.py runs
VARIANTS = [("auto", "1"), ("auto", "0"), ("auto", "2"), ("resident", "1"), ("resident", "0"), ("resident", "2"), ("column", "1")]


def solve(problem, path, symmetry, **options):
    t = _native.NativeTable(problem, solve=False, path=path)
    t.set_option("symmetry", symmetry)
    for key, value in options.items():
        t.set_option(key, str(value))
    t.solve()
    out = table_solution(t)
    t.close()
    return out


@pytest.fixture(scope="module")
def coverage20_prefix():
    """First 340 columns of BASELINE configs[2] (seed 3): 40 columns of coverage ramp, 300 at 2^20 bipartitions."""
    p = synthetic_block(n_variants=200000, coverage=20, seed=3, n_columns_limit=340)
    return p, table_solution(oracle.OracleTable(p))


@pytest.mark.parametrize("path,symmetry", VARIANTS)
def test_coverage_20_prefix_300_full_width_columns_vs_oracle(coverage20_prefix, path, symmetry):
    p, want = coverage20_prefix
    got = solve(p, path, symmetry)
    assert got == want, first_difference(want, got)


@pytest.mark.parametrize("kw", [dict(n_variants=200000, coverage=21, seed=51, n_columns_limit=66),
                                dict(n_variants=200000, coverage=23, seed=52, n_columns_limit=68)], ids=str)
def test_coverage_21_and_23_prefixes_vs_oracle(kw):
    p = synthetic_block(**kw)
    want = table_solution(oracle.OracleTable(p))
    for path, symmetry in (("auto", "1"), ("auto", "0"), ("resident", "1")):
        got = solve(p, path, symmetry)
        assert got == want, (path, symmetry, first_difference(want, got))


@pytest.mark.parametrize("kw", [dict(n_variants=200000, coverage=20, seed=3),      # BASELINE configs[2]
                                dict(n_variants=100000, coverage=20, seed=100)],   # first block of configs[4]
                         ids=["config3_200k_cov20", "config5_block_100k_cov20"])
def test_full_size_every_path_and_symmetry_setting_agree(kw):
    p = synthetic_block(**kw)
    base = solve(p, "auto", "1")
    assert wmec_cost_of_partitioning(p, np.asarray(base["partitioning"], dtype=np.uint8)) == base["cost"]
    for path, symmetry in VARIANTS[1:]:
        got = solve(p, path, symmetry)
        assert got == base, (path, symmetry, first_difference(base, got))
    # windowed solve (arena capped at a fraction of the ~13 / ~6.5 GB the records need): same answer
    got = solve(p, "auto", "1", arena_limit_bytes=3 << 30)
    assert got == base, ("windowed", first_difference(base, got))


# ---- windowed solve: the backtrace arena capped far below what the table's records need (tables beyond HBM)
WINDOW_CASES = [
    # kwargs of synthetic_block, path, arena limit in bytes
    (dict(n_variants=3000, coverage=12, seed=11), "auto", 1 << 17),
    (dict(n_variants=3000, coverage=12, seed=11), "resident", 1 << 17),
    (dict(n_variants=3000, coverage=12, seed=11), "column", 1 << 18),
    (dict(n_variants=20000, coverage=16, seed=12), "auto", 64 << 20),
]


@pytest.mark.parametrize("kw,path,limit", WINDOW_CASES, ids=str)
def test_windowed_solve_equals_the_unrestricted_solve(kw, path, limit):
    p = synthetic_block(**kw)
    t = _native.NativeTable(p, solve=False, path=path)
    t.solve()
    base = table_solution(t)
    base_launches = t.stats()["forward_launches"]
    t.close()
    t = _native.NativeTable(p, solve=False, path=path)
    t.set_option("arena_limit_bytes", str(limit))
    t.solve()
    got = table_solution(t)
    launches = t.stats()["forward_launches"]
    t.solve()                       # a second solve of the same table replays the same program
    again = table_solution(t)
    t.close()
    assert got == base, first_difference(base, got)
    assert again == base
    assert launches > base_launches, "the arena limit did not force a windowed solve"


def test_windowed_trio_equals_the_unrestricted_solve():
    p = synthetic_block(n_variants=1500, coverage=5, seed=21, trio=True)
    base = solve(p, "auto", "1")
    for path in ("auto", "resident"):
        for limit in (1 << 20, 1 << 17):
            got = solve(p, path, "1", arena_limit_bytes=limit)
            assert got == base, (path, limit, first_difference(base, got))


def test_windowed_table_of_many_connected_components():
    """Connected components are jobs of their own in an unrestricted solve; a windowed solve runs them as one job."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from gpu_multiblock import chromosome

    whole = chromosome(12, 11, seed=7, max_len=150)
    want = table_solution(oracle.OracleTable(whole))
    for limit in (1 << 22, 1 << 16):
        got = solve(whole, "auto", "1", arena_limit_bytes=limit)
        assert got == want, (limit, first_difference(want, got))


def test_unit_larger_than_the_arena_is_unsupported():
    p = synthetic_block(n_variants=400, coverage=12, seed=14)
    t = _native.NativeTable(p, solve=False)
    t.set_option("arena_limit_bytes", "256")
    with pytest.raises(_native.SolverError) as e:
        t.solve()
    assert e.value.status == _native.WHAMD_ERR_UNSUPPORTED
    t.close()


def test_full_size_trio_chunked_backtrace_equals_the_sequential_walk(monkeypatch):
    """BASELINE configs[3] (trio, 100 000 columns, coverage 15) at full size: the chunked speculative backtrace (8 orientations
    per chunk) and the sequential walk (WHAMD_BT_SEQUENTIAL, read when the table is uploaded) return the same cost, index path,
    transmission vector, partitioning and superreads; so does a windowed solve."""
    p = synthetic_block(n_variants=100000, coverage=15, seed=4, trio=True)
    chunked = solve(p, "auto", "1")
    monkeypatch.setenv("WHAMD_BT_SEQUENTIAL", "1")
    sequential = solve(p, "auto", "1")
    monkeypatch.delenv("WHAMD_BT_SEQUENTIAL")
    assert chunked == sequential, first_difference(sequential, chunked)
    windowed = solve(p, "auto", "1", arena_limit_bytes=1 << 30)
    assert windowed == sequential, first_difference(sequential, windowed)


def test_trio_with_many_recombination_events_chunked_vs_sequential_vs_oracle(monkeypatch):
    """The benchmark's trio never recombines (constant transmission vector).  With a recombination cost of 1 the optimal path
    switches transmission values all the time: the packed (index, transmission value) states, the transmission flips of the
    orientations and the argj chain of the chunk walker are all exercised, against the sequential walk and the oracle."""
    base = synthetic_block(n_variants=1400, coverage=12, seed=9, trio=True, error_rate=0.12)
    p = _native.ProblemArrays(base.read_ptr, base.var_position, base.var_allele, base.var_quality, base.read_sample_id, base.individual_id,
                              base.triple_ids, base.genotype.reshape(3, -1), None, np.ones_like(base.recombcost), base.positions, False,
                              n_variants=base.n_variants)
    want = table_solution(oracle.OracleTable(p))
    assert len(set(want["transmission"])) > 1
    switches = int(np.count_nonzero(np.diff(np.asarray(want["transmission"]))))
    assert switches > 20, switches
    chunked = solve(p, "auto", "1")
    monkeypatch.setenv("WHAMD_BT_SEQUENTIAL", "1")
    sequential = solve(p, "auto", "1")
    monkeypatch.delenv("WHAMD_BT_SEQUENTIAL")
    assert sequential == want, first_difference(want, sequential)
    assert chunked == want, first_difference(want, chunked)


@pytest.mark.parametrize("kw", [dict(n_variants=1200, coverage=9, seed=61, trio=True, distrust_genotypes=True),
                                dict(n_variants=1500, coverage=12, seed=62, trio=True, step=3),
                                dict(n_variants=900, coverage=6, seed=63, trio=True, error_rate=0.2, drop_rate=0.3)], ids=str)
def test_trio_tables_long_enough_for_chunks_vs_oracle(kw):
    """Trio tables of several dozen runs (the chunked backtrace with its eight orientations is on) against the oracle:
    distrusted genotypes (more terms than registers: the pool path), a step-3 read layout, noisy data with many BLANK entries."""
    p = synthetic_block(**kw)
    want = table_solution(oracle.OracleTable(p))
    for path in ("auto", "resident"):   # pedigree slot runs (where the cost forms fit), LDS-resident trio runs
        got = solve(p, path, "1")
        assert got == want, (path, first_difference(want, got))


def test_long_tie_heavy_table_chunked_backtrace_vs_oracle():
    """VERDICT r2 #8a: the chunked speculative backtrace against the ORACLE (not against the repo's own sequential walk)
    on a table long enough for > 150 chunks, with two-valued weights (nearly every minimum is a tie, so the minimum of an
    exit column is attained by many states and guesses do miss: the walk-again-from-the-true-state branch runs)."""
    b = synthetic_block(n_variants=64000, coverage=12, seed=77)
    p = _native.ProblemArrays(b.read_ptr, b.var_position, b.var_allele, (1 + (b.var_quality % 2)).astype(np.uint32), b.read_sample_id, b.individual_id,
                              b.triple_ids, b.genotype.reshape(1, -1), None, b.recombcost, b.positions, False, n_variants=b.n_variants)
    want = table_solution(oracle.OracleTable(p))
    t = _native.NativeTable(p)
    got = table_solution(t)
    stats = t.stats()
    t.close()
    assert stats["bt_chunks"] >= 150, stats
    assert got == want, first_difference(want, got)
    assert stats["bt_missed"] > 0 and stats["bt_rewalked"] >= stats["bt_missed"], stats   # the miss branch was exercised


def test_irregular_layout_chunked_backtrace_vs_oracle():
    """The irregular read layout of the bench (Poisson starts, geometric lengths) at a coverage the oracle affords: columns in which
    four and more reads end lie inside runs, the table is long enough for tens of backtrace chunks, two-valued weights make ties."""
    from whatshap_amd.synthetic import irregular_block

    b = irregular_block(12000, 13, seed=7)
    p = _native.ProblemArrays(b.read_ptr, b.var_position, b.var_allele, (1 + (b.var_quality % 2)).astype(np.uint32), b.read_sample_id, b.individual_id,
                              b.triple_ids, b.genotype.reshape(1, -1), None, b.recombcost, b.positions, False, n_variants=b.n_variants)
    summary = _native.plan_summary(p)
    assert summary["n_resident_columns"] > 0.99 * p.n_variants, summary   # (three ending reads per column was the limit: 2.6 % outside runs)
    want = table_solution(oracle.OracleTable(p))
    t = _native.NativeTable(p)
    got = table_solution(t)
    stats = t.stats()
    t.close()
    assert stats["bt_chunks"] >= 20, stats
    assert got == want, first_difference(want, got)
