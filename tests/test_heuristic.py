"""CPU: PedMecHeuristic.  The solver exists ONCE (whatshap_amd/csrc/heuristic_core.h); here it runs through
its single-threaded host instantiation (whamd_debug_pedmec_heuristic_create_host, a diagnostic) against (i) the compiled
reference's PedMecHeuristic (oracle/_ref) on random tie-heavy instances and synthetic blocks, (ii) the committed golden
vectors.  Everything the reference's getters return is compared: bipartition, transmission vector, haplotypes, mutations
(and the score, which the reference leaves at 0).  The device instantiation is compared the same way in test_gpu_heuristic.py."""
import pytest

import oracle
from helpers import load_golden, problem_from_json
from heuristic_cases import random_cases, result_tuple, synthetic_cases
from whatshap_amd import _native


def host(problem, row_limit):
    return result_tuple(_native.pedmec_heuristic(problem, row_limit=row_limit, host_diagnostic=True))


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref not built (no reference tree here)")
def test_random_instances_vs_the_compiled_reference():
    for name, problem, row_limit in random_cases(4711, 60):
        want = oracle.heuristic_tuple(oracle.ReferenceHeuristic(problem, row_limit=row_limit))
        assert host(problem, row_limit) == want, name


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref not built (no reference tree here)")
@pytest.mark.parametrize("case", synthetic_cases(), ids=lambda c: c[0])
def test_synthetic_blocks_vs_the_compiled_reference(case):
    name, problem, row_limit = case
    want = oracle.heuristic_tuple(oracle.ReferenceHeuristic(problem, row_limit=row_limit))
    got = _native.pedmec_heuristic(problem, row_limit=row_limit, host_diagnostic=True)
    assert result_tuple(got) == want, name
    assert 1 <= got["stats"]["max_solutions"] and got["stats"]["n_columns"] == problem.n_variants


def test_golden_vectors():
    records = load_golden("heuristic_cases.json")
    assert len(records) >= 70
    for rec in records:
        assert host(problem_from_json(rec["problem"]), rec["row_limit"]) == rec["solution"], rec["name"]


def test_input_checks_and_the_drop_in_class_without_a_device():
    import numpy as np

    from whatshap_amd.synthetic import synthetic_block

    p = synthetic_block(n_variants=40, coverage=6, seed=1, trio=True)
    short = _native.ProblemArrays(p.read_ptr, p.var_position, p.var_allele, p.var_quality, p.read_sample_id, p.individual_id, p.triple_ids,
                                  p.genotype.reshape(3, -1), None, p.recombcost[:-1], p.positions, False, n_variants=p.n_variants)
    with pytest.raises(_native.SolverError, match="one entry per position"):
        _native.pedmec_heuristic(short, host_diagnostic=True)
    order = np.arange(p.n_reads)[::-1]   # reads in descending order of their first position
    lengths = np.diff(p.read_ptr).astype(np.int64)
    idx = np.concatenate([np.arange(p.read_ptr[r], p.read_ptr[r + 1]) for r in order]).astype(np.int64)
    unsorted = _native.ProblemArrays(np.concatenate([[0], np.cumsum(lengths[order])]), p.var_position[idx], p.var_allele[idx], p.var_quality[idx],
                                     p.read_sample_id[order], p.individual_id, p.triple_ids, p.genotype.reshape(3, -1), None, p.recombcost, p.positions,
                                     False, n_variants=p.n_variants)
    with pytest.raises(_native.SolverError, match="not sorted"):
        _native.pedmec_heuristic(unsorted, host_diagnostic=True)
    if _native.device_count() == 0:   # the product entry point needs the GPU: no CPU fallback
        with pytest.raises(_native.SolverError) as e:
            _native.pedmec_heuristic(p)
        assert e.value.status == _native.WHAMD_ERR_DEVICE and "no CPU fallback" in str(e.value)
