"""CPU: the slot-run planner (whatshap_amd/csrc/slot_plan.cpp) checked against the oracle without a GPU.
`whamd_debug_emulate_slot_plan` executes the plan cell by cell the way the kernels do (physical cell indices, per-slot
deltas, decision bits incl. the mirror-image ones, record layout, entry / exit layouts, backtrace blobs); cost and index
path must equal the oracle's, for every preferred slice size and with / without the complement symmetry."""
import os
import random
import sys

import numpy as np
import pytest

import oracle
from whatshap_amd import _native
from whatshap_amd.synthetic import random_small_instance, synthetic_block

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def agrees(problem, **kw):
    o = oracle.OracleTable(problem)
    want_idx, _ = o.index_path()
    idx, score, run_columns = _native.emulate_slot_plan(problem, o.n_columns, **kw)
    return score == o.optimal_score() and bool((idx == want_idx).all()), run_columns


def test_random_tie_heavy_instances():
    rng = random.Random(11)
    with_runs = 0
    for i in range(250):
        p = random_small_instance(rng, mode="single", allow_conflict=False)
        ok, run_columns = agrees(p, slot_r=1 + i % 3)
        assert ok, i
        with_runs += run_columns > 0
    assert with_runs > 150


@pytest.mark.parametrize("seed", range(6))
def test_cost_variants_slice_sizes_and_symmetry(seed):
    from test_gpu_parity import _irregular_problem, _variant_of

    rng = np.random.default_rng(seed)
    base = synthetic_block(n_variants=120, coverage=10 + seed % 4, seed=seed, step=1 + seed % 3)
    n = base.n_variants
    variants = {
        "plain": base,
        "ties": _variant_of(base, quality=(1 + (base.var_quality % 2)).astype(np.uint32)),
        "homozygous": _variant_of(base, genotype=rng.choice([0, 1, 1, 2], size=(1, n)).astype(np.uint8)),
        "distrust": _variant_of(base, gl=rng.integers(0, 40, size=(1, n, 3)).astype(np.float64), distrust=True),
        "heavy": _variant_of(base, quality=base.var_quality * np.uint32(450)),
        "irregular": _irregular_problem(seed, 160, False, 13),
    }
    for name, p in variants.items():
        for slot_l, symmetry, slot_r in ((9, 1, 3), (10, 1, 2), (11, 0, 3), (9, 0, 2), (8, 1, 2), (10, 1, 1), (8, 0, 1)):
            ok, run_columns = agrees(p, slot_l=slot_l, symmetry=symmetry, slot_r=slot_r)
            assert ok, (name, slot_l, symmetry, slot_r)
            assert run_columns > 0.8 * p.n_variants or name == "irregular", (name, run_columns)


def test_connected_components_and_pedigrees():
    from dist_worker import multi_block_instance

    for seed in range(20):
        ok, _ = agrees(multi_block_instance(seed))
        assert ok, seed
    trio = synthetic_block(n_variants=50, coverage=6, seed=1, trio=True)
    with pytest.raises(_native.SolverError) as e:
        _native.emulate_slot_plan(trio, 50)
    assert e.value.status == _native.WHAMD_ERR_UNSUPPORTED


def _reads_problem(reads, n_variants, seed):
    """A single-individual problem from explicit reads [(first variant, last variant)], alleles / qualities seeded."""
    rng = np.random.default_rng(seed)
    reads = sorted(reads)
    ptr, pos, allele, qual = [0], [], [], []
    for first, last in reads:
        for v in range(first, last + 1):
            pos.append(100 * (v + 1)); allele.append(int(rng.integers(0, 2))); qual.append(int(rng.integers(1, 4)))
        ptr.append(len(pos))
    return _native.ProblemArrays(np.array(ptr, dtype=np.uint64), np.array(pos, dtype=np.int32), np.array(allele, dtype=np.uint8), np.array(qual, dtype=np.uint32),
                                 np.zeros(len(reads), dtype=np.int32), np.array([0], dtype=np.uint32), np.zeros(0, dtype=np.uint32),
                                 np.ones((1, n_variants), dtype=np.uint8), None, np.zeros(n_variants, dtype=np.uint32),
                                 np.array([100 * (v + 1) for v in range(n_variants)], dtype=np.uint32), False, n_variants=n_variants)


@pytest.mark.parametrize("seed", range(4))
def test_many_reads_ending_in_one_column(seed):
    """Four to eight reads whose last variant is the same column (an irregular layout does that in every tenth run): the run
    continues through it (SLOT_MAXEND = 8; the third and later ending reads come out of the row's second line)."""
    rng = np.random.default_rng(40 + seed)
    n = 40
    reads = []
    for stop in (9, 17, 26, 33):           # columns where 4 + seed reads end at once
        for q in range(4 + seed):
            reads.append((int(stop - 2 - rng.integers(0, 5)), stop))
    for first in range(0, n - 6, 3):       # a background of staggered reads
        reads.append((first, min(n - 1, first + int(rng.integers(5, 12)))))
    p = _reads_problem(reads, n, seed)
    for slot_l, slot_r in ((10, 2), (9, 2), (11, 3), (9, 1)):
        ok, run_columns = agrees(p, slot_l=slot_l, slot_r=slot_r)
        assert ok, (seed, slot_l, slot_r)
        assert run_columns >= n - 2, run_columns
