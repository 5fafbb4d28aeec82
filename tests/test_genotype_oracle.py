"""The CPU restatement of GenotypeDPTable (oracle/genotype_oracle.py) against the REAL reference class
(whatshap.core.GenotypeDPTable built into oracle/_ref/cy): single individual, trio, quartet; uniform and random priors;
BLANK entries; columns without reads."""
import numpy as np
import pytest

from genotype_cases import random_case, reference_likelihoods
from oracle import genotype_oracle
from refobjects import reference_core


@pytest.mark.parametrize("mode,seeds", [("single", range(12)), ("trio", range(8)), ("quartet", range(4))])
def test_restatement_equals_the_reference_class(mode, seeds):
    ref = reference_core()
    for seed in seeds:
        p = random_case(100 * len(mode) + seed, n_variants=7, n_reads=9 if mode == "single" else 7, max_len=4, mode=mode,
                        max_coverage=6 if mode == "single" else 4, uniform_prior=seed % 3 == 0)
        want = reference_likelihoods(p, ref)
        got = np.asarray(genotype_oracle.genotype_likelihoods(p), dtype=np.float64)
        assert got.shape == want.shape
        assert np.allclose(got, want, rtol=1e-12, atol=1e-15), (mode, seed, np.abs(got - want).max())
        assert np.allclose(got.sum(axis=2), 1.0)


def _matrix_problem(lines, quality=10):
    """Reads given as strings (one character per variant, ' ' = not covered), uniform priors, recombination cost 1:
    check_genotyping_single_individual of the reference's tests/test_genotyping.py:66-95 with scaling 10."""
    from whatshap_amd import _native

    read_ptr, pos, alle = [0], [], []
    for line in lines:
        for col, ch in enumerate(line):
            if ch != " ":
                pos.append(10 * (col + 1)); alle.append(int(ch))
        read_ptr.append(len(pos))
    n_var = len(set(pos))
    return _native.ProblemArrays(
        np.asarray(read_ptr, dtype=np.uint64), np.asarray(pos, dtype=np.int32), np.asarray(alle, dtype=np.uint8),
        np.full(len(pos), quality, dtype=np.uint32), np.zeros(len(lines), dtype=np.int32), np.asarray([0], dtype=np.uint32),
        np.zeros(0, dtype=np.uint32), np.ones((1, n_var), dtype=np.uint8), np.full((1, n_var, 3), 1.0 / 3.0),
        np.ones(n_var, dtype=np.uint32), None, False, n_variants=n_var)


REFERENCE_VECTORS = [
    # tests/test_genotyping.py:108-123 (test_geno_exact1), :125-141 (exact2), :144-155 (exact3)
    (["11", " 01"], [[0.06666666666666667, 0.3333333333333333, 0.6], [0.20930232558139536, 0.5813953488372093, 0.20930232558139536],
                     [0.06666666666666667, 0.3333333333333333, 0.6]]),
    (["11", "11"], [[0.00914139256727894, 0.25040580948312685, 0.7404527979495942]] * 2),
    (["01", "11"], [[0.22163406214039125, 0.5567318757192175, 0.22163406214039125],
                    [0.009896432681242807, 0.18849252013808976, 0.8016110471806674]]),
]


@pytest.mark.parametrize("lines,want", REFERENCE_VECTORS, ids=["exact1", "exact2", "exact3"])
def test_known_answers_of_the_reference_tests(lines, want):
    got = np.asarray(genotype_oracle.genotype_likelihoods(_matrix_problem(lines)), dtype=np.float64)[0]
    assert np.allclose(got, want, rtol=1e-9)


def golden_cases():
    from helpers import load_golden, problem_from_json

    for case in load_golden("genotype_cases.json")["cases"]:
        yield case["name"], problem_from_json(case["problem"]), np.asarray([[[float(x) for x in col] for col in ind] for ind in case["likelihoods"]])


def test_restatement_equals_the_committed_golden_vectors():
    """tests/golden/genotype_cases.json (generated from the reference class by make_genotype_golden.py): usable where
    neither /root/reference nor the built reference modules exist."""
    n = 0
    for name, problem, want in golden_cases():
        got = np.asarray(genotype_oracle.genotype_likelihoods(problem), dtype=np.float64)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-15), name
        n += 1
    assert n == 6


def test_device_path_fails_loudly_without_a_gpu_and_checks_its_input_first():
    """No CPU fallback: with valid input and no visible device the C ABI reports WHAMD_ERR_DEVICE; missing priors are an input
    error before any device is touched (the reference asserts them, src/transitionprobabilitycomputer.cpp:66)."""
    from whatshap_amd import _native

    p = _matrix_problem(["11", " 01"])
    if _native.device_count() == 0:
        with pytest.raises(_native.SolverError) as e:
            _native.genotype_likelihoods(p, 3)
        assert e.value.status == _native.WHAMD_ERR_DEVICE and "no CPU fallback" in str(e.value)
    no_priors = _native.ProblemArrays(p.read_ptr, p.var_position, p.var_allele, p.var_quality, p.read_sample_id, p.individual_id, p.triple_ids,
                                      p.genotype.reshape(1, -1), None, p.recombcost, None, False, n_variants=p.n_variants)
    with pytest.raises(_native.SolverError) as e:
        _native.genotype_likelihoods(no_priors, 3)
    assert e.value.status == _native.WHAMD_ERR_INVALID and "priors" in str(e.value)
