"""GPU (-m gpu): the pedigree slot runs (kernels_pedslots.h; T = 4 and T = 16) through the C ABI against the oracle --
synthetic quartets and double trios at coverage 9-11 over several hundred columns, Mendelian-consistent mixed genotypes,
tie-heavy weights, many recombination events, irregular read layouts, the windowed solve, and the older paths they
replaced (LDS-resident trio runs, per-column kernels) on the same inputs.  Bit-exact (integer path)."""
import numpy as np
import pytest

import oracle
from helpers import first_difference, native_solution, table_solution
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block

pytestmark = pytest.mark.gpu


def _with(p, quality=None, recomb=None):
    return _native.ProblemArrays(p.read_ptr, p.var_position, p.var_allele, p.var_quality if quality is None else quality, p.read_sample_id,
                                 p.individual_id, p.triple_ids, p.genotype.reshape(p.n_individuals, p.n_variants), None,
                                 p.recombcost if recomb is None else recomb, p.positions, False, n_variants=p.n_variants)


def solve(problem, path="auto", **options):
    t = _native.NativeTable(problem, solve=False, path=path)
    for k, v in options.items():
        t.set_option(k, str(v))
    t.solve()
    out = table_solution(t)
    stats = t.stats()
    t.close()
    return out, stats


QUARTETS = [
    dict(n_variants=320, coverage=9, seed=71, quartet=True),
    dict(n_variants=300, coverage=10, seed=72, quartet=True, mixed_genotypes=True),
    dict(n_variants=300, coverage=11, seed=73, quartet=True, step=1),
    dict(n_variants=400, coverage=8, seed=74, quartet=True, error_rate=0.15, drop_rate=0.3),
    dict(n_variants=300, coverage=9, seed=75, two_trios=True),
    dict(n_variants=300, coverage=10, seed=76, two_trios=True, mixed_genotypes=True),
]


@pytest.mark.parametrize("kw", QUARTETS, ids=str)
def test_quartets_and_double_trios_vs_oracle(kw):
    """T = 16 at coverage >= 9 over >= 300 columns (the table is long enough for the chunked backtrace with its
    orientations): pedigree slot runs == oracle == per-column kernels, with far fewer launches."""
    p = synthetic_block(**kw)
    want = table_solution(oracle.OracleTable(p))
    got, stats = solve(p)
    assert got == want, first_difference(want, got)
    col, col_stats = solve(p, "column")
    assert col == want, first_difference(want, col)
    assert stats["forward_launches"] * 2 < col_stats["forward_launches"], (stats["forward_launches"], col_stats["forward_launches"])


@pytest.mark.parametrize("kw", [dict(n_variants=500, coverage=10, seed=81, quartet=True), dict(n_variants=600, coverage=11, seed=82, trio=True),
                                dict(n_variants=400, coverage=9, seed=83, two_trios=True, mixed_genotypes=True)], ids=str)
def test_tie_heavy_weights_and_cheap_recombination(kw):
    """Two-valued weights and recombination costs of 0 / 1 / 2: nearly every minimum is a tie (Gray-rank rule between the
    cells of a pair, lowest previous transmission value in the butterfly) and the optimal path switches transmission values."""
    b = synthetic_block(**kw)
    p = _with(b, quality=(1 + (b.var_quality % 2)).astype(np.uint32), recomb=(b.recombcost % 3).astype(np.uint32))
    want = table_solution(oracle.OracleTable(p))
    got, _ = solve(p)
    assert got == want, first_difference(want, got)
    if kw.get("trio"):
        res, _ = solve(p, "resident")
        assert res == want, first_difference(want, res)


def test_many_recombination_events_in_a_quartet():
    b = synthetic_block(n_variants=700, coverage=10, seed=85, quartet=True, error_rate=0.12)
    p = _with(b, recomb=np.ones_like(b.recombcost))
    want = table_solution(oracle.OracleTable(p))
    assert len(set(want["transmission"])) > 2
    got, _ = solve(p)
    assert got == want, first_difference(want, got)


@pytest.mark.parametrize("slot_l", [4, 5, 6])
def test_fewer_local_slots(slot_l):
    """Narrower workgroups (1 - 4 waves): more grid slots, other exchange layouts, other wave-slot / lane-slot splits."""
    p = synthetic_block(n_variants=260, coverage=10, seed=86, trio=True, mixed_genotypes=True)
    want = table_solution(oracle.OracleTable(p))
    got, _ = solve(p, slot_l=slot_l)
    assert got == want, first_difference(want, got)


def test_windowed_quartet_equals_the_unrestricted_solve():
    p = synthetic_block(n_variants=1500, coverage=7, seed=87, quartet=True)
    base, st0 = solve(p)
    for limit in (1 << 21, 1 << 18):
        got, st = solve(p, arena_limit_bytes=limit)
        assert got == base, (limit, first_difference(base, got))
        assert st["forward_launches"] > st0["forward_launches"], "the arena limit did not force a windowed solve"


@pytest.mark.parametrize("kw", [dict(n_variants=400, coverage=9, seed=88, trio=True, distrust_genotypes=True),
                                dict(n_variants=700, coverage=12, seed=89, trio=True, distrust_genotypes=True, mixed_genotypes=True),
                                # full width: BASELINE configs[3]'s ReadSet with the genotypes not trusted, >= 25 launches of 256 workgroups
                                dict(n_variants=100000, coverage=15, seed=4, trio=True, distrust_genotypes=True, n_columns_limit=430)], ids=str)
def test_untrusted_genotypes_on_pedigree_runs_vs_oracle(kw):
    """Up to 15 cost forms per transmission value (src/pedigreecolumncostcomputer.cpp:14-50,101-114): pedslot_run<2, 16> == oracle ==
    the LDS-resident trio runs it replaces in the default dispatch == the per-column kernels."""
    p = synthetic_block(**kw)
    want = table_solution(oracle.OracleTable(p))
    got, stats = solve(p)
    assert got == want, first_difference(want, got)
    assert stats["forward_launches"] <= p.n_variants // 6, "the table did not run on pedigree slot runs"
    old, _ = solve(p, "resident")
    assert old == want, first_difference(want, old)
    if p.n_variants <= 700:
        col, _ = solve(p, "column")
        assert col == want, first_difference(want, col)


def test_full_size_trio_old_and_new_runs_agree():
    """BASELINE configs[3] at full size: pedigree slot runs == LDS-resident trio runs on every output."""
    p = synthetic_block(n_variants=100000, coverage=15, seed=4, trio=True)
    new, st_new = solve(p)
    old, st_old = solve(p, "resident")
    assert new == old, first_difference(old, new)
    assert st_new["total_ms"] < st_old["total_ms"]


@pytest.mark.parametrize("coverage", [3, 4, 5, 6, 7, 8, 10])
def test_factorised_lines_in_workgroups_of_every_size(coverage, monkeypatch):
    """An untrusted trio at coverage 3 ... 10: workgroups of 64 ... 512 threads.  The constants of the factorised lines (slots.h PSLOT_FACT, the K table)
    are staged by however many threads the workgroup has; sixteen forms (WHAMD_NO_PED_FACT) give the same tuple; both == the oracle."""
    p = synthetic_block(n_variants=90, coverage=coverage, seed=300 + coverage, trio=True, distrust_genotypes=True, mixed_genotypes=coverage % 2 == 0)
    want = table_solution(oracle.OracleTable(p))
    assert _native.plan_summary(p)["n_fact_runs"] > 0
    got, _ = solve(p)
    assert got == want, first_difference(want, got)
    monkeypatch.setenv("WHAMD_NO_PED_FACT", "1")
    assert _native.plan_summary(p)["n_fact_runs"] == 0
    generic, _ = solve(p)
    assert generic == want, first_difference(want, generic)


@pytest.mark.parametrize("mode", ["three_children", "three_trios", "big_family"])
def test_pedigrees_beyond_two_trios_and_six_individuals_vs_oracle(mode):
    """Three trios (T = 64: a family with three children, three unrelated trios) and seven individuals in one table: the
    generic per-column kernel (column_step_wide) -- the device path refuses a pedigree only beyond 3 trios / 12 individuals."""
    import random

    from whatshap_amd.synthetic import random_small_instance

    rng = random.Random({"three_children": 191, "three_trios": 192, "big_family": 193}[mode])
    compared = 0
    for _ in range(50):
        p = random_small_instance(rng, mode=mode, max_variants=8, max_reads=7)
        try:
            want = table_solution(oracle.OracleTable(p))
        except oracle.OracleError as e:
            with pytest.raises(_native.SolverError, match="Mendelian"):
                solve(p)
            assert "Mendelian" in str(e)
            continue
        got, _ = solve(p)   # (untrusted genotypes with 7+ individuals: more cost terms per column than the templated kernels stage -- generic kernel)
        assert got == want, (mode, first_difference(want, got))
        compared += 1
    assert compared > 30


@pytest.mark.parametrize("seed", range(4))
def test_four_reads_ending_in_one_column_of_a_pedigree_run(seed):
    """The fourth ending read of a column (PSLOT_MAXEND = 4) comes out of the row with scalar loads; its decision is bit 7 of the
    lane's record byte, its slot sits next to the count in the backtrace column."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_pedslot_plan import _trio_reads_problem

    rng = np.random.default_rng(60 + seed)
    n = 36
    reads = []
    for stop in (8, 15, 23, 30):
        for q in range(4):
            reads.append((int(stop - 2 - rng.integers(0, 4)), stop))
    for first in range(0, n - 6, 4):
        reads.append((first, min(n - 1, first + int(rng.integers(5, 10)))))
    p = _trio_reads_problem(reads, n, seed)
    want = table_solution(oracle.OracleTable(p))
    assert _native.plan_summary(p)["n_resident_columns"] >= n - 5
    for path in ("auto", "resident", "column"):
        got, _ = solve(p, path)
        assert got == want, (path, first_difference(want, got))


FULL_WIDTH = [
    # BASELINE configs[3] itself (trio, coverage 15, seed 4): 30 columns of ramp, then > 40 full-width pedslot_run launches of 14 columns
    # (256 workgroups x 8 waves, 3 wave slots + grid slots), long enough for the chunked backtrace with its eight orientations
    ("trio coverage 15", dict(n_variants=100000, coverage=15, seed=4, trio=True, n_columns_limit=640)),
    # the bench's quartet (two trios sharing parents, T = 16, coverage 13, seed 5)
    ("quartet coverage 13", dict(n_variants=50000, coverage=13, seed=5, quartet=True, n_columns_limit=280)),
]


@pytest.mark.parametrize("name,kw", FULL_WIDTH, ids=[n for n, _ in FULL_WIDTH])
@pytest.mark.parametrize("ties", [False, True], ids=["weights as generated", "two-valued weights"])
def test_full_width_pedigree_runs_vs_oracle(name, kw, ties):
    """The pedigree configurations of the bench line at FULL width against the oracle (the reference does ~100 columns/s on this
    shape): lowest-j rule of the min-plus step (src/pedigreedptable.cpp:264-300) and the Gray-order rule of the projection (:306-327),
    on the slot runs (`auto`) and on the per-column kernels."""
    p = synthetic_block(**kw)
    if ties:
        rng = np.random.default_rng(7)
        p = _with(p, quality=rng.choice(np.array([4, 8], dtype=np.uint32), size=p.var_quality.size),
                  recomb=rng.choice(np.array([0, 1, 2, 12], dtype=np.uint32), size=p.recombcost.size))
    want = table_solution(oracle.OracleTable(p))
    got, stats = solve(p)
    assert got == want, (name, first_difference(want, got))
    assert stats["forward_launches"] <= p.n_variants // 8, "the table did not run on pedigree slot runs"
    if not ties and name.startswith("trio"):
        assert stats["bt_chunks"] >= 2, "the table did not go through the chunked backtrace"
    col, _ = solve(p, "column")
    assert col == want, (name, "column", first_difference(want, col))


@pytest.mark.parametrize("kw,recomb", [(dict(n_variants=300, coverage=10, seed=611, trio=True), 1 << 20),
                                      (dict(n_variants=300, coverage=9, seed=612, quartet=True), 1 << 18),
                                      (dict(n_variants=260, coverage=9, seed=613, quartet=True, distrust_genotypes=True), 1 << 18)], ids=str)
def test_tables_whose_values_leave_no_room_for_packed_keys_take_the_staged_step(kw, recomb):
    """Round 5: the min-plus step over the previous transmission value runs on packed keys value << TB | j where every finite value of the table stays below
    2^(31 - TB) - 1 (slot_plan.cpp, SlotRun::yflags bit 4); beyond that -- here: recombination costs of 2^20 (trio) / 2^18 (quartet) per column, an upper bound
    of the table's values above 2^29 / 2^27 but below the 2^30 the slot runs need -- the staged comparison of rounds 3 - 4 runs (the other template
    instantiation).  Both against the oracle; the same tables with cheap recombination are what every other pedigree test solves."""
    p = synthetic_block(**kw)
    rng = np.random.default_rng(kw["seed"])
    rc = rng.choice(np.array([recomb, recomb + 1, recomb // 2], dtype=np.uint32), size=p.recombcost.size)
    q = _native.ProblemArrays(p.read_ptr, p.var_position, p.var_allele, p.var_quality, p.read_sample_id, p.individual_id, p.triple_ids,
                              p.genotype.reshape(p.n_individuals, p.n_variants), p.genotype_likelihoods, rc, p.positions, p.distrust_genotypes, n_variants=p.n_variants)
    want = table_solution(oracle.OracleTable(q))
    got, stats = solve(q)
    assert got == want, first_difference(want, got)
    assert stats["forward_launches"] <= p.n_variants // 3, "the table did not run on pedigree slot runs"
