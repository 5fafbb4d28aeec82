"""The known-answer cases of the reference's own unit tests for the wMEC/PedMEC path, as data.

Sources (whatshap/whatshap @ 2025-07-11):
  tests/test_phasing.py:154-238      eight read matrices + one weight matrix, each solved 4 ways
                                     (single individual / trio with two read-less individuals) x
                                     (all heterozygous / distrust_genotypes with flat likelihoods)
  tests/test_pedigreephasing.py      trio / quartet / double-trio cases with exact costs
  tests/test_verification.py:24-43   the string case and tests/test.matrix

Each case is a function returning ``Case`` (inputs through the mirror classes of whatshap_amd.core plus the
expectations the reference test asserts).  The golden fixtures in tests/golden/ hold the outputs of the
compiled reference for exactly these inputs (tests/golden/make_golden.py).
"""

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

from helpers import biallelic_gt, biallelic_gt_list, matrix_to_readset, string_to_readset, string_to_readset_pedigree
from whatshap_amd.core import NumericSampleIds, Pedigree, PhredGenotypeLikelihoods, ReadSet


@dataclass
class Case:
    name: str
    readset: ReadSet
    recombcost: List[int]
    pedigree: Pedigree
    distrust_genotypes: bool = False
    positions: Optional[List[int]] = None
    expected_cost: Optional[int] = None
    expected_haplotypes: Optional[List[Tuple[str, str]]] = None  # per individual, order-free
    allowed_transmission: Optional[List[List[int]]] = None
    constant_transmission: bool = False
    all_heterozygous: Optional[bool] = None  # single-individual cases: which brute-force model applies
    weights: Optional[str] = None


PHASING_MATRICES: Dict[str, Tuple[str, Optional[str]]] = {
    "trivial": ("""
          11
           01
        """, None),
    "phase1": ("""
     10
     010
     010
    """, None),
    "phase2": ("""
      1  11010
      00 00101
      001 0101
    """, None),
    "phase3": ("""
      1  11010
      00 00101
      001 01010
    """, None),
    "phase4": ("""
      1  11010
      00 00101
      001 01110
       1    111
    """, None),
    "phase5": ("""
      0             0
      110111111111
      00100
           0001000000
           000
            10100
                  101
    """, None),
    "weighted1": ("""
      1  11010
      00 00101
      001 01110
       1    111
    """, """
      2  13112
      11 23359
      223 56789
       2    111
    """),
}

MATRIX_LINES = [
    "1 2 1011", "2 3 1001", "3 3 011", "4 3 011", "5 4 0011", "6 6 00", "7 6 00", "8 6 11", "9 7 01", "10 8 11110",
]  # tests/test.matrix


def _single(readset, all_heterozygous, n_individuals):
    """check_phasing_single_individual (tests/test_phasing.py:79-151): recombination cost 1, every genotype 0/1,
    flat likelihoods [0,0,0] when genotypes are distrusted."""
    positions = readset.get_positions()
    pedigree = Pedigree(NumericSampleIds())
    gls = [None if all_heterozygous else PhredGenotypeLikelihoods([0, 0, 0])] * len(positions)
    for i in range(n_individuals):
        pedigree.add_individual(f"individual{i}", [biallelic_gt(1) for _ in positions], gls)
    if n_individuals == 3:
        pedigree.add_relationship("individual0", "individual1", "individual2")
    return [1] * len(positions), pedigree


def phasing_cases() -> List[Case]:
    cases = []
    for name, (reads, weights) in PHASING_MATRICES.items():
        for n_ind in (1, 3):
            for all_het in (False, True):
                rs = string_to_readset(reads, weights)
                recomb, ped = _single(rs, all_het, n_ind)
                cases.append(Case(f"phasing_{name}_{'trio' if n_ind == 3 else 'single'}_{'het' if all_het else 'distrust'}",
                                  rs, recomb, ped, distrust_genotypes=not all_het, constant_transmission=True,
                                  all_heterozygous=all_het, weights=weights))
    return cases


def verification_cases() -> List[Case]:
    cases = []
    for all_het in (True, False):
        rs = string_to_readset(PHASING_MATRICES["phase5"][0])
        recomb, ped = _single(rs, all_het, 1)
        cases.append(Case(f"verify_string_{'het' if all_het else 'distrust'}", rs, recomb, ped, distrust_genotypes=not all_het,
                          all_heterozygous=all_het))
        rs = matrix_to_readset(MATRIX_LINES)
        recomb, ped = _single(rs, all_het, 1)
        cases.append(Case(f"verify_matrix_{'het' if all_het else 'distrust'}", rs, recomb, ped, distrust_genotypes=not all_het,
                          all_heterozygous=all_het))
    return cases


def _pedigree(genotypes: Sequence[Sequence[int]], relationships, likelihoods=None, names=None) -> Pedigree:
    ped = Pedigree(NumericSampleIds())
    for i, g in enumerate(genotypes):
        name = names[i] if names else f"individual{i}"
        ped.add_individual(name, biallelic_gt_list(g), None if likelihoods is None else likelihoods[i])
    for f, m, c in relationships:
        ped.add_relationship(f, m, c)
    return ped


TRIO = [("individual0", "individual1", "individual2")]
QUARTET = TRIO + [("individual0", "individual1", "individual3")]


def pedigree_cases() -> List[Case]:
    c = []
    c.append(Case("trio1", string_to_readset_pedigree("""
      A 111
      A 010
      A 110
      B 001
      B 110
      B 101
      C 001
      C 010
      C 010
    """), [10, 10, 10], _pedigree([[1, 2, 1], [1, 1, 1], [0, 1, 1]], TRIO), expected_cost=2, constant_transmission=True,
                  expected_haplotypes=[("111", "010"), ("001", "110"), ("010", "001")]))
    c.append(Case("trio2", string_to_readset_pedigree("""
      A 00
      A 00
      B 11
      B 11
      C 11
      C 00
    """), [10, 10, 10], _pedigree([[2, 2], [0, 0], [1, 1]], TRIO), expected_cost=8, constant_transmission=True,
                  expected_haplotypes=[("11", "11"), ("00", "00"), ("00", "11")]))
    c.append(Case("trio3", string_to_readset_pedigree("""
      A 1111
      B 1010
      C 111000
      C 010101
      B 0101
      A  0000
      B  1010
      C  1010
      C  1100
      A   0000
      A   1111
      B   1010
      B    010
    """), [3, 3, 3, 4, 3, 3], _pedigree([[1] * 6, [1] * 6, [1, 2, 1, 1, 0, 1]], TRIO), expected_cost=4,
                  allowed_transmission=[[0, 0, 0, 1, 1, 1], [1, 1, 1, 0, 0, 0], [2, 2, 2, 3, 3, 3], [3, 3, 3, 2, 2, 2]],
                  expected_haplotypes=[("111111", "000000"), ("010101", "101010"), ("111000", "010101")]))
    trio45 = """
      B 101
      B 101
      B 101
      A 111
      A 111
      A 111
      C 111
      C 111
      C 111
    """
    c.append(Case("trio4", string_to_readset_pedigree(trio45), [1, 1, 1], _pedigree([[1] * 3] * 3, TRIO), expected_cost=2,
                  allowed_transmission=[[0, 2, 0], [2, 0, 2], [1, 3, 1], [3, 1, 3]],
                  expected_haplotypes=[("111", "000"), ("101", "010"), ("111", "000")]))
    c.append(Case("trio5", string_to_readset_pedigree(trio45), [2, 2, 2], _pedigree([[1] * 3] * 3, TRIO), expected_cost=3,
                  constant_transmission=True, expected_haplotypes=[("111", "000"), ("111", "000"), ("111", "000")]))
    # the reference test passes 3 recombination costs for 4 columns (reads recombcost[3] out of bounds);
    # the C ABI pads with the last value -- the golden fixture uses the explicit 4-entry list
    c.append(Case("trio_pure_genetic", string_to_readset_pedigree(""), [2, 2, 2, 2],
                  _pedigree([[2, 1, 1, 0], [1, 2, 2, 1], [1, 1, 1, 0]], TRIO), positions=[10, 20, 30, 40], expected_cost=0,
                  constant_transmission=True, expected_haplotypes=[("1110", "1000"), ("1111", "0110"), ("1000", "0110")]))
    c.append(Case("doubletrio_pure_genetic", string_to_readset_pedigree(""), [2, 2, 2, 2],
                  _pedigree([[1, 2, 1, 0], [1, 0, 1, 1], [2, 1, 1, 0], [1, 2, 2, 1], [1, 1, 1, 0]],
                            [("individualA", "individualB", "individualC"), ("individualC", "individualD", "individualE")],
                            names=["individualA", "individualB", "individualC", "individualD", "individualE"]),
                  positions=[10, 20, 30, 40], expected_cost=0, constant_transmission=True,
                  expected_haplotypes=[("0100", "1110"), ("0011", "1000"), ("1110", "1000"), ("1111", "0110"), ("1000", "0110")]))
    c.append(Case("quartet1", string_to_readset_pedigree("""
      A 111
      A 010
      A 110
      B 001
      B 110
      B 101
      C 001
      C 010
      C 010
      D 001
      D 010
      D 010
    """), [10, 10, 10], _pedigree([[1, 2, 1], [1, 1, 1], [0, 1, 1], [0, 1, 1]], QUARTET), expected_cost=2,
                  constant_transmission=True,
                  expected_haplotypes=[("111", "010"), ("001", "110"), ("001", "010"), ("001", "010")]))
    c.append(Case("quartet2", string_to_readset_pedigree("""
      A 111111
      A 000000
      B 010101
      B 101010
      C 000000
      C 010101
      D 000000
      D 010101
    """), [3] * 6, _pedigree([[1] * 6, [1] * 6, [0, 1, 0, 1, 0, 1], [0, 1, 0, 1, 0, 1]], QUARTET), expected_cost=0,
                  constant_transmission=True,
                  expected_haplotypes=[("111111", "000000"), ("010101", "101010"), ("000000", "010101"), ("000000", "010101")]))
    c.append(Case("quartet3", string_to_readset_pedigree("""
      A 1111
      A 0000
      B 1010
      C 111000
      C 010101
      D 000000
      D 010
      B 0101
      C  1100
      D  10010
      A   0000
      A   1111
      B   1010
      B   0101
    """), [3, 3, 3, 4, 3, 3], _pedigree([[1] * 6, [1] * 6, [1, 2, 1, 1, 0, 1], [0, 1, 0, 0, 1, 0]], QUARTET), expected_cost=8,
                  expected_haplotypes=[("111111", "000000"), ("010101", "101010"), ("111000", "010101"), ("000000", "010010")]))
    flat = [PhredGenotypeLikelihoods([0, 0, 0])] * 3
    mother = [PhredGenotypeLikelihoods([0, 0, 0]), PhredGenotypeLikelihoods([0, 0, 1]), PhredGenotypeLikelihoods([5, 0, 5])]
    c.append(Case("trio_genotype_likelihoods", string_to_readset_pedigree("""
      A 111
      A 010
      A 110
      B 001
      B 110
      B 101
      C 001
      C 010
      C 010
    """), [10, 10, 10], _pedigree([[0, 0, 0]] * 3, TRIO, likelihoods=[mother, flat, flat]), distrust_genotypes=True,
                  expected_cost=3, constant_transmission=True,
                  expected_haplotypes=[("111", "010"), ("001", "110"), ("001", "010")]))
    return c


def all_cases() -> List[Case]:
    return phasing_cases() + verification_cases() + pedigree_cases()
