"""CPU: the oracle restatement against the golden vectors of the compiled reference, against the
reference's own known answers, against brute force, and (where oracle/_ref exists) against the
compiled reference on fresh random instances."""
import itertools
import random

import pytest

import oracle
from helpers import first_difference, load_golden, problem_from_json, table_solution
from reference_cases import all_cases
from whatshap_amd.core import problem_from_objects
from whatshap_amd.synthetic import random_small_instance, synthetic_block


def oracle_outcome(problem):
    try:
        return table_solution(oracle.OracleTable(problem)), None
    except oracle.OracleError as e:
        return None, str(e)


@pytest.mark.parametrize("fixture", ["reference_cases.json", "random_tie_heavy.json", "synthetic_small.json"])
def test_oracle_matches_golden(fixture):
    records = load_golden(fixture)
    assert records
    n_errors = 0
    for rec in records:
        solution, error = oracle_outcome(problem_from_json(rec["problem"]))
        assert error == rec["error"], rec["name"]
        if error is None:
            assert solution == rec["solution"], f"{rec['name']}: {first_difference(rec['solution'], solution)}"
        else:
            n_errors += 1
    if fixture == "random_tie_heavy.json":
        assert n_errors > 0  # the generator injects Mendelian conflicts; both sides must raise on the same inputs


def brute_force(readset, all_heterozygous):
    """min over all bipartitions of sum over columns of the cheapest allele pair (tests/testhelpers.py:125-177 idea)."""
    reads = list(readset)
    positions = readset.get_positions()
    pairs = [(0, 1), (1, 0)] if all_heterozygous else [(0, 0), (0, 1), (1, 0), (1, 1)]
    best, count = None, 0
    for partition in range(2 ** len(reads)):
        cost = 0
        for p in positions:
            side = [[], []]
            for n, read in enumerate(reads):
                for v in read:
                    if v.position == p:
                        side[(partition >> n) & 1].append(v)
            cost += min(sum(v.quality for v in side[0] if v.allele != a0) + sum(v.quality for v in side[1] if v.allele != a1)
                        for a0, a1 in pairs)
        if best is None or cost < best:
            best, count = cost, 1
        elif cost == best:
            count += 1
    return best, count


@pytest.mark.parametrize("case", all_cases(), ids=lambda c: c.name)
def test_reference_known_answers(case):
    problem = problem_from_objects(case.readset, case.recombcost, case.pedigree, case.distrust_genotypes, case.positions)
    sol, err = oracle_outcome(problem)
    assert err is None
    if case.expected_cost is not None:
        assert sol["cost"] == case.expected_cost
    if case.all_heterozygous is not None and len(case.readset) < 10:
        expected, _ = brute_force(case.readset, case.all_heterozygous)
        assert sol["cost"] == expected
    if case.constant_transmission:
        assert len(set(sol["transmission"])) <= 1
    if case.allowed_transmission is not None:
        assert sol["transmission"] in case.allowed_transmission
    if case.expected_haplotypes is not None:
        for ind, expected in enumerate(case.expected_haplotypes):
            got = tuple(sorted(["".join(map(str, sol["allele0"][ind])), "".join(map(str, sol["allele1"][ind]))]))
            assert got == tuple(sorted(expected)), (case.name, ind)


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref not built (no reference tree here)")
def test_oracle_vs_compiled_reference_random():
    rng = random.Random(4242)
    for i in range(300):
        p = random_small_instance(rng)
        want, werr = None, None
        try:
            want = table_solution(oracle.ReferenceTable(p))
        except oracle.OracleError as e:
            werr = str(e)
        got, gerr = oracle_outcome(p)
        assert gerr == werr, i
        assert got == want, f"{i}: {first_difference(want, got)}"


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref not built (no reference tree here)")
@pytest.mark.parametrize("mode", ["three_children", "three_trios", "big_family"])
def test_oracle_vs_compiled_reference_larger_pedigrees(mode):
    """Three trios (T = 64) and a seven-individual family: the restatement is general in T and in the number of individuals,
    like the reference (src/pedigreepartitions.cpp:7-42); pinned against the compiled reference before the device path is."""
    if not oracle.have_reference():
        pytest.skip("compiled reference not built")
    rng = random.Random({"three_children": 91, "three_trios": 92, "big_family": 93}[mode])
    compared = 0
    for _ in range(60):
        p = random_small_instance(rng, mode=mode, max_variants=7, max_reads=6)
        try:
            want = table_solution(oracle.ReferenceTable(p))
        except oracle.OracleError as e:
            with pytest.raises(oracle.OracleError, match=str(e)):
                oracle.OracleTable(p)
            continue
        assert table_solution(oracle.OracleTable(p)) == want
        compared += 1
    assert compared > 40


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref not built (no reference tree here)")
@pytest.mark.parametrize("kw", [dict(n_variants=300, coverage=10, seed=31), dict(n_variants=150, coverage=9, seed=32, trio=True),
                                dict(n_variants=3000, coverage=14, seed=33, n_columns_limit=40)], ids=str)
def test_oracle_vs_compiled_reference_synthetic(kw):
    p = synthetic_block(**kw)
    assert table_solution(oracle.OracleTable(p)) == table_solution(oracle.ReferenceTable(p))


def _with_blank_alleles(p, rng):
    """Copy of `p` in which some interior variants of reads carry Entry::BLANK (allele 2) with a non-zero phred."""
    import numpy as np
    from whatshap_amd import _native

    alle = p.var_allele.copy()
    ptr = p.read_ptr.astype(np.int64)
    for r in range(p.n_reads):
        for i in range(int(ptr[r]) + 1, int(ptr[r + 1]) - 1):
            if rng.random() < 0.3:
                alle[i] = 2
    gl = None if p.genotype_likelihoods is None else p.genotype_likelihoods.reshape(p.n_individuals, p.n_variants, 3)
    return _native.ProblemArrays(p.read_ptr, p.var_position, alle, p.var_quality, p.read_sample_id, p.individual_id, p.triple_ids,
                                 p.genotype.reshape(p.n_individuals, p.n_variants), gl, p.recombcost, p.positions,
                                 p.distrust_genotypes, n_variants=p.n_variants)


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref not built (no reference tree here)")
def test_blank_alleles_inside_reads_vs_compiled_reference():
    """Entry::BLANK (2) as the allele of a read's own variant is legal input for the reference: set/update_partitioning
    skip it (src/pedigreecolumncostcomputer.cpp:69-70, 93-94).  The restatement must treat it the same way (its phred
    is ignored); allele 3 (EQUAL_SCORES) is rejected (the reference's assert(false), :71-72)."""
    import numpy as np
    from whatshap_amd import _native

    rng = random.Random(99)
    n_with_blank = 0
    for i in range(120):
        p = _with_blank_alleles(random_small_instance(rng, allow_conflict=False, max_variants=9, max_reads=7), rng)
        n_with_blank += int((p.var_allele == 2).any())
        want = table_solution(oracle.ReferenceTable(p))
        got, err = oracle_outcome(p)
        assert err is None and got == want, f"{i}: {first_difference(want, got)}"
    assert n_with_blank > 40
    bad = _native.ProblemArrays([0, 2], [10, 20], [0, 3], [1, 1], [0], [0], [], np.ones((1, 2)), None, [1, 1], None, False)
    _, err = oracle_outcome(bad)
    assert err is not None and "allele" in err
