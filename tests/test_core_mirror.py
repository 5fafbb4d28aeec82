"""CPU: container behaviour of the whatshap.core mirror (modelled on the reference's tests/test_reads.py and
tests/test_pedigree.py) and the flattening into the C-ABI views."""
import numpy as np
import pytest

from helpers import biallelic_gt, string_to_readset
from whatshap_amd.core import (Genotype, NumericSampleIds, Pedigree, PhredGenotypeLikelihoods, Read, ReadSet, Variant,
                               problem_from_objects)


def test_read_container():
    r = Read("name", 15)
    assert r.name == "name" and r.mapqs[0] == 15 and len(r) == 0
    r.add_variant(100, 1, 37)
    r.add_variant(23, 0, 99)
    assert len(r) == 2 and not r.is_sorted()
    assert r[0] == Variant(100, 1, 37) and r[-1] == Variant(23, 0, 99)
    assert 23 in r and 24 not in r
    r.sort()
    assert [v.position for v in r] == [23, 100] and r.is_sorted()
    r[0] = Variant(24, 1, 5)
    assert r[0].quality == 5
    with pytest.raises(IndexError):
        r[2]
    r.add_variant(24, 0, 1)
    with pytest.raises(RuntimeError, match="Duplicate variant"):
        r.sort()


def test_readset_add_copies_and_rejects_duplicates():
    rs = ReadSet()
    r = Read("r", 1, 0, 0)
    r.add_variant(10, 0, 1)
    rs.add(r)
    r.add_variant(20, 1, 1)
    assert len(rs[0]) == 1  # the set holds a copy, as the reference does (core.pyx:281-286)
    with pytest.raises(RuntimeError, match="duplicate read name"):
        rs.add(r)
    assert rs[(0, "r")] is rs[0]
    with pytest.raises(KeyError):
        rs[(1, "r")]


def test_readset_sort_by_first_position_then_hash():
    rs = ReadSet()
    for name, pos in (("c", 30), ("a", 10), ("b", 10), ("d", 20)):
        r = Read(name, 1, 0, 0)
        r.add_variant(pos, 0, 1)
        r.add_variant(pos + 5, 0, 1)
        rs.add(r)
    rs.sort()
    firsts = [r[0].position for r in rs]
    assert firsts == sorted(firsts)
    from whatshap_amd._native import read_sort_hash
    tied = [r.name for r in rs if r[0].position == 10]
    assert tied == sorted(tied, key=lambda n: read_sort_hash(n, 0))
    assert rs.get_positions() == [10, 15, 20, 25, 30, 35]
    sub = rs.subset([2, 0])
    assert [r.name for r in sub] == [rs[0].name, rs[2].name]


def test_genotype_and_likelihoods():
    assert Genotype([1, 0]) == Genotype([0, 1]) and Genotype([1, 0]).get_index() == 1
    assert Genotype([1, 1]).get_index() == 2 and Genotype([1, 1]).is_homozygous()
    assert Genotype([]).is_none() and not Genotype([0, 2]).is_diploid_and_biallelic()
    assert str(Genotype([1, 0])) == "0/1"
    gl = PhredGenotypeLikelihoods([0, 5, 7])
    assert gl[Genotype([0, 1])] == 5 and list(gl) == [0, 5, 7]
    with pytest.raises(RuntimeError):
        PhredGenotypeLikelihoods([0, 1])


def test_pedigree_and_flattening():
    ids = NumericSampleIds()
    ped = Pedigree(ids)
    ped.add_individual("mom", [biallelic_gt(1), biallelic_gt(2)], [PhredGenotypeLikelihoods([1, 2, 3]), None])
    ped.add_individual("dad", [biallelic_gt(0), Genotype([0, 2])])
    ped.add_individual("kid", [biallelic_gt(1), biallelic_gt(1)])
    ped.add_relationship("dad", "mom", "kid")
    assert len(ped) == 3 and ped.variant_count == 2
    assert ped.genotype("mom", 1) == Genotype([1, 1]) and ped.genotype_likelihoods("mom", 1) is None
    rs = string_to_readset("""
      10
      01
    """, sample_ids=[ids["mom"], ids["kid"]])
    p = problem_from_objects(rs, [0, 3], ped, False, None)
    assert p.read_ptr.tolist() == [0, 2, 4] and p.var_position.tolist() == [10, 20, 10, 20]
    assert p.genotype.reshape(3, 2).tolist() == [[1, 2], [0, 255], [1, 1]]
    assert p.triple_ids.tolist() == [ids["dad"], ids["mom"], ids["kid"]]
    assert np.isnan(p.genotype_likelihoods.reshape(3, 2, 3)[0, 1]).all()
    assert p.genotype_likelihoods.reshape(3, 2, 3)[0, 0].tolist() == [1, 2, 3]
