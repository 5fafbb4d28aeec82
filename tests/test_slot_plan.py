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
