"""GPU (-m gpu): several tables as ONE sequence of launches (whamd_dptable_enqueue_many -> DeviceTable::enqueue_group,
slot_group / pedslot_group): every table of a batched solve must give exactly what it gives alone -- cost, index path,
transmission vector, partitioning, superreads with qualities -- and what the oracle gives.

Independent tables are what `whatshap phase` produces per chromosome x family (whatshap/cli/phase.py:467,486,604); the tie rules
being pinned are src/pedigreedptable.cpp:264-300 (lowest j) and :306-327 (Gray order).
"""
import numpy as np
import pytest

import oracle
from helpers import first_difference, table_solution
from whatshap_amd import _native
from whatshap_amd.synthetic import irregular_block, synthetic_block

pytestmark = pytest.mark.gpu


def two_valued(problem, seed):
    """Qualities from {5, 10}: nearly every minimum is attained more than once (the tie rules carry the result)."""
    rng = np.random.default_rng(seed)
    q = rng.choice(np.array([5, 10], dtype=np.uint32), size=problem.var_quality.size)
    return _native.ProblemArrays(problem.read_ptr, problem.var_position, problem.var_allele, q, problem.read_sample_id, problem.individual_id,
                                 problem.triple_ids, problem.genotype, problem.genotype_likelihoods, problem.recombcost, problem.positions,
                                 problem.distrust_genotypes, n_variants=problem.n_variants)


def the_24_tables():
    """24 DIFFERENT tables: widths from 1 to 16 workgroups, lengths from 300 to 4000 columns (some longer than two backtrace chunks, so
    their runs leave speculative seeds inside the group kernel), regular and irregular layouts, tie-heavy weights, trios, a quartet."""
    out = []
    for i, (cov, n) in enumerate([(15, 4000), (15, 3000), (14, 2500), (13, 1800), (12, 900), (12, 400), (11, 350), (10, 300)]):
        out.append(("single", synthetic_block(n_variants=n, coverage=cov, seed=40 + i)))
    for i, (cov, n) in enumerate([(15, 3000), (14, 2000), (12, 600), (12, 450)]):
        out.append(("irregular", irregular_block(n, cov, seed=60 + i)))
    for i, (cov, n) in enumerate([(14, 2200), (12, 500), (11, 400)]):
        out.append(("ties", two_valued(synthetic_block(n_variants=n, coverage=cov, seed=70 + i), 170 + i)))
    for i, (cov, n) in enumerate([(13, 2500), (12, 1500), (10, 420), (9, 380), (9, 300)]):
        out.append(("trio", synthetic_block(n_variants=n, coverage=cov, seed=80 + i, trio=True)))
    out.append(("trio ties", two_valued(synthetic_block(n_variants=400, coverage=9, seed=88, trio=True), 188)))
    out.append(("quartet", synthetic_block(n_variants=1200, coverage=11, seed=90, quartet=True)))
    out.append(("quartet", synthetic_block(n_variants=300, coverage=8, seed=91, quartet=True)))
    out.append(("single", synthetic_block(n_variants=700, coverage=9, seed=92)))
    assert len(out) == 24
    return out


ORACLE_ON = range(24)   # (the CPU restatement needs ~25 s for all of them)


@pytest.fixture(scope="module")
def tables_and_alone():
    cases = the_24_tables()
    alone = []
    for _, p in cases:
        t = _native.NativeTable(p)
        alone.append(table_solution(t))
        t.close()
    return cases, alone


def test_24_different_tables_batched_equal_one_by_one_and_the_oracle(tables_and_alone):
    cases, alone = tables_and_alone
    tables = [_native.NativeTable(p, solve=False) for _, p in cases]
    _native.enqueue_many(tables)
    for t in tables:
        t.wait()
    for i, t in enumerate(tables):
        got = table_solution(t)
        assert got == alone[i], (i, cases[i][0], first_difference(alone[i], got))
    assert any(t.stats()["bt_chunks"] > 0 for t in tables), "no table of the batch went through the chunked backtrace"
    assert all(t.stats()["group_tables"] == 24 for t in tables), "the batch did not share its launches"
    for i in ORACLE_ON:
        want = table_solution(oracle.OracleTable(cases[i][1]))
        assert alone[i] == want, (i, cases[i][0], first_difference(want, alone[i]))
    # a second solve of the same batch (buffers re-armed on the group's stream) and of a reversed, smaller batch
    _native.enqueue_many(tables)
    for t in tables:
        t.wait()
    for i, t in enumerate(tables):
        assert table_solution(t) == alone[i], (i, "second batched solve")
    some = [tables[i] for i in (23, 17, 9, 0)]
    _native.enqueue_many(some)
    for t in some:
        t.wait()
    for t, i in zip(some, (23, 17, 9, 0)):
        assert table_solution(t) == alone[i], (i, "reordered batch")
    # ... and a table of the batch solved alone afterwards
    tables[3].solve()
    assert table_solution(tables[3]) == alone[3]
    for t in tables:
        t.close()


def test_batch_with_tables_outside_the_group(tables_and_alone):
    """Per-column and LDS-resident tables keep their own streams next to a group; a table with connected components brings several
    runs per super-step into the group launch."""
    cases, alone = tables_and_alone
    picks = (4, 6, 17, 22)
    tables = [_native.NativeTable(cases[i][1], solve=False, path=("column" if i == 6 else ("resident" if i == 17 else None))) for i in picks]
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from gpu_multiblock import chromosome

    comp = chromosome(9, 11, seed=6, max_len=150)
    comp_alone = table_solution(_native.NativeTable(comp))
    tables.append(_native.NativeTable(comp, solve=False))
    _native.enqueue_many(tables)
    for t in tables:
        t.wait()
    for t, i in zip(tables, picks):
        assert table_solution(t) == alone[i], (i, first_difference(alone[i], table_solution(t)))
    assert table_solution(tables[-1]) == comp_alone
    assert comp_alone == table_solution(oracle.OracleTable(comp))
    for t in tables:
        t.close()


def test_full_width_tables_batched(monkeypatch):
    """Three coverage-20 tables (256 workgroups each) in one launch per super-step: more workgroups than the chip holds at once
    (enqueue_many would keep so few wide tables on their own streams: WHAMD_GROUP_ALWAYS forces the shared launches)."""
    monkeypatch.setenv("WHAMD_GROUP_ALWAYS", "1")
    ps = [synthetic_block(n_variants=200000, coverage=20, seed=3 + i, n_columns_limit=1500 + 200 * i) for i in range(3)]
    alone = []
    for p in ps:
        t = _native.NativeTable(p)
        alone.append(table_solution(t))
        t.close()
    tables = [_native.NativeTable(p, solve=False) for p in ps]
    _native.enqueue_many(tables)
    for t in tables:
        t.wait()
    for t, a in zip(tables, alone):
        assert t.stats()["group_tables"] == 3
        assert table_solution(t) == a, first_difference(a, table_solution(t))
        t.close()


def test_enqueue_many_rejects_a_table_listed_twice_and_takes_an_empty_list():
    p = synthetic_block(n_variants=300, coverage=9, seed=1)
    a, b = _native.NativeTable(p, solve=False), _native.NativeTable(p, solve=False)
    with pytest.raises(_native.SolverError) as e:
        _native.enqueue_many([a, b, a])
    assert e.value.status == _native.WHAMD_ERR_INVALID
    _native.enqueue_many([])
    _native.wait_many([])
    _native.enqueue_many([a, b])   # the refused call left nothing in flight
    _native.wait_many([a, b])
    assert table_solution(a) == table_solution(b) == table_solution(oracle.OracleTable(p))
    a.close(); b.close()


def test_shared_launches_layout_eight_cells_per_thread_vs_oracle(monkeypatch):
    """Tables created with the option shared_launches = 1 (whamd_dptable_create_with_options; what blocks.solve_blocks passes for windows
    of more than four tables): wide single-individual tables take eight cells per thread and twelve local slots (Y form with Kr[0..7]), narrow
    ones and pedigrees keep their layout.  Every table equals the oracle -- alone and sharing launches with the others."""
    cases = [synthetic_block(n_variants=200000, coverage=20, seed=3, n_columns_limit=420),       # 300+ full-width columns of BASELINE configs[2]
             synthetic_block(n_variants=100000, coverage=20, seed=100, n_columns_limit=900),     # a configs[4] block, several backtrace chunks
             two_valued(synthetic_block(n_variants=1200, coverage=19, seed=7), 77),              # tie-heavy
             irregular_block(1500, 20, seed=9),
             synthetic_block(n_variants=2000, coverage=18, seed=8),
             synthetic_block(n_variants=3000, coverage=15, seed=5),                              # narrow: keeps four cells per thread
             synthetic_block(n_variants=600, coverage=12, seed=6, trio=True)]
    want = [table_solution(oracle.OracleTable(p)) for p in cases]
    opts = {"shared_launches": "1"}
    for p, w in zip(cases, want):   # alone
        t = _native.NativeTable(p, options=opts)
        assert table_solution(t) == w, first_difference(w, table_solution(t))
        t.close()
    tables = [_native.NativeTable(p, solve=False, options=opts) for p in cases]
    _native.enqueue_many(tables)
    _native.wait_many(tables)
    for t, w in zip(tables, want):
        assert t.stats()["group_tables"] == len(cases)
        assert table_solution(t) == w, first_difference(w, table_solution(t))
    launches = [t.stats()["forward_launches"] for t in tables]
    plain = _native.NativeTable(cases[0], solve=False)
    plain.solve()
    assert launches[0] <= plain.stats()["forward_launches"], "twelve local slots did not give longer runs"
    plain.close()
    for t in tables:
        t.close()
    # the explicit options, and eleven local slots (256-thread workgroups)
    for extra in ({"slot_r": "3", "slot_l": "12"}, {"slot_r": "3", "slot_l": "11"}, {"slot_r": "3", "slot_l": "10", "symmetry": "0"}):
        t = _native.NativeTable(cases[0], options=extra)
        assert table_solution(t) == want[0], (extra, first_difference(want[0], table_solution(t)))
        t.close()



def test_96_tables_in_one_group_vs_single_solves_and_the_oracle():
    """a cohort's worth of tables per launch.  96 tables (more than three workgroups per CU in a launch, the X kernel's streamed variant
    slot_groupx for the single individuals, slot_group for what it does not take, pedslot_group for the trios): every one equals its solve alone; ten of
    them -- each kind, each layout -- equal the oracle."""
    cases = []
    for i in range(72):
        cov, n = (15, 1300) if i % 3 == 0 else ((14, 900) if i % 3 == 1 else (12, 500))
        kind = i % 6
        if kind == 4:
            cases.append(("irregular", irregular_block(n, cov, seed=300 + i)))
        elif kind == 5:
            cases.append(("ties", two_valued(synthetic_block(n_variants=n, coverage=cov, seed=300 + i), 400 + i)))
        else:
            cases.append(("single", synthetic_block(n_variants=n + 7 * i, coverage=cov, seed=300 + i)))
    for i in range(24):
        cases.append(("trio", synthetic_block(n_variants=400 + 20 * i, coverage=9 + i % 4, seed=500 + i, trio=True)))
    assert len(cases) == 96
    tables = [_native.NativeTable(p, solve=False, options={"shared_launches": "1"}) for _, p in cases]
    _native.enqueue_many(tables)
    _native.wait_many(tables)
    assert max(t.stats()["group_tables"] for t in tables) >= 72
    batched = [table_solution(t) for t in tables]
    for t in tables:
        t.close()
    for k, (kind, p) in enumerate(cases):
        alone = _native.NativeTable(p)
        mine = table_solution(alone)
        alone.close()
        assert batched[k] == mine, (k, kind, first_difference(batched[k], mine))
    for k in (0, 1, 2, 4, 5, 10, 11, 40, 72, 95):
        want = table_solution(oracle.OracleTable(cases[k][1]))
        assert batched[k] == want, (k, cases[k][0], first_difference(batched[k], want))


def test_superreads_made_on_the_device_equal_the_hosts_loop(monkeypatch):
    """A single-individual table with trusted genotypes gets its superreads from superreads_single / superreads_group (kernels_backtrace.h) and its
    partitioning from the column each read enters in; the debug library's WHAMD_HOST_SUPERREADS=1 sends the same tables through the host's loop
    (problem.cpp finish_columns).  Alone and as a group, tie-heavy weights and homozygous genotypes included; everything compared, qualities too."""
    problems = [synthetic_block(n_variants=n, coverage=cov, seed=300 + i) for i, (cov, n) in enumerate([(15, 3000), (12, 700), (9, 400), (5, 50), (15, 21000)])]
    problems.append(irregular_block(1500, 14, seed=310))
    problems.append(two_valued(synthetic_block(n_variants=900, coverage=12, seed=311), 411))
    homo = synthetic_block(n_variants=800, coverage=11, seed=312)
    g = homo.genotype.copy()
    g[::3] = 0
    g[1::7] = 2
    problems.append(_native.ProblemArrays(homo.read_ptr, homo.var_position, homo.var_allele, homo.var_quality, homo.read_sample_id, homo.individual_id, homo.triple_ids, g,
                                          homo.genotype_likelihoods, homo.recombcost, homo.positions, homo.distrust_genotypes, n_variants=homo.n_variants))

    def solve_all():
        alone = []
        for p in problems:
            t = _native.NativeTable(p)
            alone.append(table_solution(t))
            t.close()
        tables = [_native.NativeTable(p, solve=False) for p in problems]
        _native.enqueue_many(tables)
        _native.wait_many(tables)
        grouped = [table_solution(t) for t in tables]
        for t in tables:
            t.close()
        return alone, grouped

    device_alone, device_grouped = solve_all()
    saved = _native._lib
    try:
        monkeypatch.setenv("WHAMD_HOST_SUPERREADS", "1")
        _native.use_debug_library()
        host_alone, host_grouped = solve_all()
    finally:
        _native._lib = saved
    for i in range(len(problems)):
        assert first_difference(device_alone[i], host_alone[i]) is None, (i, first_difference(device_alone[i], host_alone[i]))
        assert first_difference(device_grouped[i], host_grouped[i]) is None, (i, first_difference(device_grouped[i], host_grouped[i]))
        assert first_difference(device_alone[i], device_grouped[i]) is None, (i, first_difference(device_alone[i], device_grouped[i]))
    for i in (1, 2, 3, 5, 6, 7):
        want = table_solution(oracle.OracleTable(problems[i]))
        assert first_difference(device_alone[i], want) is None, (i, first_difference(device_alone[i], want))
