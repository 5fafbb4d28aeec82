"""CPU: the host planner of the forward pass (resident runs / folded columns / per-column steps) -- no device needed.
Checks the invariants the kernels rely on and that the expected schedule comes out for the benchmark shapes."""
import random

import numpy as np
import pytest

from whatshap_amd import _native
from whatshap_amd.synthetic import random_small_instance, synthetic_block


def test_benchmark_shape_is_scheduled_as_slot_runs():
    """Default path of a single individual: register-resident slot runs (slots.h)."""
    p = synthetic_block(n_variants=3000, coverage=20, seed=3)
    s = _native.plan_summary(p)
    assert s["invariants_ok"] == 1 and s["n_columns"] == 3000 and s["max_coverage"] == 20
    assert s["n_resident_columns"] >= 2980          # everything but the ramp's odd columns and the last column
    assert s["max_workgroups"] == 256               # 11 local slots, 9 grid slots, half of the workgroups launched
    assert s["n_halved_runs"] >= 0.9 * s["n_runs"]
    assert 15 <= s["n_resident_columns"] / s["n_runs"] <= 64
    assert s["max_lds_bytes"] <= 64 * 1024
    assert s["n_steps"] < s["n_columns"] / 10
    # one byte per thread and ending read: far below the LDS-resident runs' records
    assert s["backtrace_bytes"] <= _native.plan_summary(p, "resident")["backtrace_bytes"]


def test_benchmark_shape_is_scheduled_as_resident_runs():
    p = synthetic_block(n_variants=3000, coverage=20, seed=3)
    s = _native.plan_summary(p, "resident")
    assert s["invariants_ok"] == 1 and s["n_columns"] == 3000 and s["max_coverage"] == 20
    assert s["n_resident_columns"] >= 2990          # everything but the last column(s)
    assert s["max_workgroups"] == 256               # 9 grid reads at coverage 20, half of the workgroups launched
    assert s["n_halved_runs"] >= 0.9 * s["n_runs"]
    assert 15 <= s["n_resident_columns"] / s["n_runs"] <= 48
    assert s["n_folded_columns"] >= 0.4 * s["n_columns"]  # every second column starts a read and ends none
    assert s["max_lds_bytes"] <= 160 * 1024
    assert s["n_steps"] < s["n_columns"] / 10


def test_coverage_23_still_runs_resident():
    p = synthetic_block(n_variants=400, coverage=23, seed=9)
    s = _native.plan_summary(p, "resident")
    assert s["invariants_ok"] == 1 and s["max_coverage"] == 23
    assert s["max_workgroups"] == 512 and s["n_resident_columns"] >= 300  # 10 grid reads, halved
    s = _native.plan_summary(p)   # slot runs: 11 local + 12 grid slots, halved
    assert s["invariants_ok"] == 1 and s["max_workgroups"] == 2048 and s["n_resident_columns"] >= 300


def test_column_path_request_has_no_runs_and_trios_get_their_own_runs():
    p = synthetic_block(n_variants=500, coverage=12, seed=5)
    s = _native.plan_summary(p, "column")
    assert s["n_runs"] == 0 and s["n_steps"] == 500 and s["invariants_ok"] == 1
    trio = synthetic_block(n_variants=600, coverage=15, seed=6, trio=True)
    s = _native.plan_summary(trio)
    assert s["invariants_ok"] == 1 and s["n_runs"] > 0 and s["n_resident_columns"] >= 580
    assert s["n_folded_columns"] == 0          # the transmission argmin is recorded on every column
    assert s["max_workgroups"] >= 32 and s["max_lds_bytes"] <= 160 * 1024
    assert s["max_workgroups"] == 256 and s["max_run_columns"] >= 12   # pedigree slot runs: 7 local + 8 grid slots
    s = _native.plan_summary(trio, "resident")   # the LDS-resident trio runs remain selectable
    assert s["invariants_ok"] == 1 and s["n_runs"] > 0 and s["n_resident_columns"] >= 580
    quartet = synthetic_block(n_variants=400, coverage=12, seed=7, quartet=True, mixed_genotypes=True)
    s = _native.plan_summary(quartet)   # two trios: pedigree slot runs with T = 16 (2 lane slots + 3 wave slots)
    assert s["invariants_ok"] == 1 and s["n_runs"] > 0 and s["n_resident_columns"] >= 300 and s["max_workgroups"] == 128
    s = _native.plan_summary(quartet, "column")
    assert s["n_runs"] == 0 and s["invariants_ok"] == 1
    distrust = synthetic_block(n_variants=300, coverage=10, seed=8, trio=True, distrust_genotypes=True)
    s = _native.plan_summary(distrust)   # 9 cost forms per transmission value: not for the pedigree slot runs
    assert s["invariants_ok"] == 1 and s["n_runs"] > 0 and s["max_workgroups"] <= 64


@pytest.mark.parametrize("seed", range(6))
def test_invariants_on_irregular_reads(seed):
    rng = np.random.default_rng(seed)
    n_var = 600
    read_ptr, pos, alle, qual = [0], [], [], []
    n_reads = 0
    cov = np.zeros(n_var, dtype=int)
    for start in range(0, n_var - 2):
        for _ in range(int(rng.integers(0, 3))):
            end = min(n_var, start + int(rng.choice([2, 3, 5, 9, 17, 30, 60])))
            if cov[start:end].max() >= 22:
                continue
            cols = [c for c in range(start, end) if c in (start, end - 1) or rng.random() < 0.7]
            cov[start:end] += 1
            for c in cols:
                pos.append(10 * (c + 1)); alle.append(int(rng.integers(0, 2))); qual.append(int(rng.integers(1, 30)))
            read_ptr.append(len(pos))
            n_reads += 1
    p = _native.ProblemArrays(read_ptr, pos, alle, qual, np.zeros(n_reads), [0], [], np.ones((1, n_var)), None, [1] * n_var,
                              [10 * (c + 1) for c in range(n_var)], False)
    for path in ("auto", "resident"):
        s = _native.plan_summary(p, path)
        assert s["invariants_ok"] == 1 and s["n_columns"] == n_var, path
        assert s["n_resident_columns"] > 0


def test_invariants_on_random_small_instances():
    rng = random.Random(5)
    checked = 0
    for _ in range(300):
        p = random_small_instance(rng, allow_conflict=False)
        s = _native.plan_summary(p)
        assert s["invariants_ok"] == 1
        assert _native.plan_summary(p, "resident")["invariants_ok"] == 1
        checked += s["n_runs"] > 0
    assert checked > 20


def test_components_are_found_and_no_run_spans_one():
    """Single individual: the planner records where connected components start (the device driver runs them as independent
    jobs) and cuts runs there; the count equals what the Python block splitter finds."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dist_worker import multi_block_instance
    from whatshap_amd.blocks import split_independent_blocks

    for seed in range(8):
        whole = multi_block_instance(seed)
        s = _native.plan_summary(whole)
        assert s["invariants_ok"] == 1
        assert s["n_components"] == len(split_independent_blocks(whole))
    single = synthetic_block(n_variants=600, coverage=12, seed=3)
    assert _native.plan_summary(single)["n_components"] == 1
    trio = synthetic_block(n_variants=300, coverage=9, seed=3, trio=True)
    assert _native.plan_summary(trio)["n_components"] == 0  # coupled through the transmission vector: never split


def _chain_problem(n_reads, edit=None, positions=None):
    """n_reads reads of three variants each, read r at variants r, r + 1, r + 2 (positions 10 * (v + 1)); `edit(ptr, pos)` may damage it."""
    import numpy as np

    ptr = np.arange(0, 3 * (n_reads + 1), 3, dtype=np.uint64)
    pos = (10 * (np.arange(n_reads)[:, None] + np.arange(3)[None, :] + 1)).astype(np.int32).reshape(-1)
    if edit:
        edit(ptr, pos)
    n_var = n_reads + 2
    return _native.ProblemArrays(ptr, pos, np.zeros(pos.size, dtype=np.uint8), np.ones(pos.size, dtype=np.uint32), np.zeros(n_reads, dtype=np.int32),
                                 np.array([0], dtype=np.uint32), np.zeros(0, dtype=np.uint32), np.ones((1, n_var), dtype=np.uint8), None,
                                 np.ones(n_var, dtype=np.uint32), positions, False, n_variants=n_var)


def test_validation_of_the_parallel_flattener_reports_the_first_failing_read():
    """build_problem validates the reads on several host threads (40 000 reads: three ranges); the error is the one the reference's
    sequential constructor would raise -- that of the FIRST failing read (src/columniterator.cpp:22-41), whichever range finds what."""
    import numpy as np

    n = 40000

    def late_unsorted_variants_early_unsorted_reads(ptr, pos):
        pos[3 * 30000 + 1] = pos[3 * 30000]          # read 30 000: variants not strictly increasing
        pos[3 * 100: 3 * 100 + 3] -= 500             # read 100 starts before read 99
    with pytest.raises(_native.SolverError, match="reads in ReadSet are not sorted"):
        _native.plan_summary(_chain_problem(n, late_unsorted_variants_early_unsorted_reads))

    def only_late(ptr, pos):
        pos[3 * 30000 + 1] = pos[3 * 30000]
    with pytest.raises(_native.SolverError, match="encountered read with unsorted variants"):
        _native.plan_summary(_chain_problem(n, only_late))

    def read_without_variants(ptr, pos):
        ptr[20001] = ptr[20000]                       # read 20 000 is empty (the next one takes its variants)
    with pytest.raises(_native.SolverError, match="No variants present"):
        _native.plan_summary(_chain_problem(n, read_without_variants))

    positions = (10 * (np.arange(n + 2) + 1)).astype(np.uint32)
    missing = np.delete(positions, 777)               # a position some read starts / ends at is not in the list
    with pytest.raises(_native.SolverError, match="not in the position list"):
        _native.plan_summary(_chain_problem(n, None, missing))
    swapped = positions.copy()
    swapped[[5000, 5001]] = swapped[[5001, 5000]]
    with pytest.raises(_native.SolverError, match="not in the position list|strictly increasing"):
        _native.plan_summary(_chain_problem(n, None, swapped))
    assert _native.plan_summary(_chain_problem(n, None, positions))["n_columns"] == n + 2
