"""readselection against the REAL reference module (whatshap.readselect, built into
oracle/_ref/cy by oracle/build_cython_ref.py): the same set of read indices, ties included.

The reference's choice among equally scored reads depends on the iteration order of Python sets and of one
std::unordered_set (whatshap/readselect.pyx:97,142,150); the native code replays those orders (csrc/readselect.cpp), and
these tests are what pins that claim: quality-free, tie-heavy inputs where almost every pick is a tie, read indices far
beyond the small-set table sizes, preferred sources, bridging on and off.
"""
import numpy as np
import pytest

from refobjects import reference_core
from whatshap_amd import _native, core
from whatshap_amd.readselect import readselection


def random_reads(rng, n_reads, n_variants, max_len, gap_prob=0.0, qualities=(10,), n_sources=1, spacing=10):
    """[(source_id, [(position, allele, quality)])] sorted the way a ReadSet is after sort()."""
    reads = []
    for _ in range(n_reads):
        length = int(rng.integers(2, max_len + 1))
        start = int(rng.integers(0, max(1, n_variants - length)))
        idx = [v for v in range(start, min(n_variants, start + length)) if rng.random() >= gap_prob]
        if len(idx) < 2:
            idx = [start, min(n_variants - 1, start + 1)] if start + 1 < n_variants else [start - 1, start]
        reads.append((int(rng.integers(0, n_sources)),
                      [(100 + spacing * v, int(rng.integers(0, 2)), int(rng.choice(qualities))) for v in idx]))
    reads.sort(key=lambda r: r[1][0][0])
    return reads


def build(reads, module):
    rs = module.ReadSet()
    for i, (source, variants) in enumerate(reads):
        r = module.Read(f"r{i}", 60, source, 0)
        for pos, allele, quality in variants:
            r.add_variant(pos, allele, quality)
        rs.add(r)
    return rs


def both(reads, max_cov, preferred=None, bridging=True):
    ref = reference_core()
    import whatshap.readselect as ref_readselect

    want = ref_readselect.readselection(build(reads, ref), max_cov, preferred, bridging)
    got = readselection(build(reads, core), max_cov, preferred, bridging)
    return set(want), got


CASES = [
    # n_reads, n_variants, max_len, gap_prob, qualities, max_cov
    (40, 30, 6, 0.0, (10,), 3),
    (200, 60, 8, 0.0, (10,), 5),
    (300, 80, 10, 0.3, (10,), 4),
    (300, 80, 10, 0.3, (5, 10, 20), 4),
    (1500, 300, 12, 0.2, (10,), 8),
    (1500, 300, 12, 0.2, (7, 30), 15),
    (3000, 200, 25, 0.4, (10,), 15),
    (800, 100, 5, 0.0, (10,), 2),
    (600, 50, 30, 0.5, (1, 2), 20),
]


@pytest.mark.parametrize("bridging", [True, False])
@pytest.mark.parametrize("case", CASES, ids=str)
def test_same_selection_as_the_reference_module(case, bridging):
    n_reads, n_variants, max_len, gap_prob, qualities, max_cov = case
    for seed in range(6):
        rng = np.random.default_rng(1000 * seed + n_reads)
        reads = random_reads(rng, n_reads, n_variants, max_len, gap_prob, qualities)
        want, got = both(reads, max_cov, None, bridging)
        assert got == want, (seed, sorted(want ^ got)[:10])


@pytest.mark.parametrize("preferred", [{0}, {1, 2}, {5}, set(), {0, 1, 2}])
def test_preferred_sources_first(preferred):
    for seed in range(5):
        rng = np.random.default_rng(77 + seed)
        reads = random_reads(rng, 700, 120, 10, 0.25, (10, 20), n_sources=3)
        want, got = both(reads, 6, preferred, True)
        assert got == want, (seed, sorted(want ^ got)[:10])


def test_many_reads_beyond_the_x4_growth_of_sets():
    """More than 50000 reads: CPython's sets switch from x4 to x2 growth there (Objects/setobject.c set_add_entry)."""
    rng = np.random.default_rng(5)
    reads = random_reads(rng, 60000, 4000, 8, 0.1, (10,))
    want, got = both(reads, 10)
    assert got == want, sorted(want ^ got)[:10]


def test_selection_respects_the_coverage_bound():
    rng = np.random.default_rng(9)
    reads = random_reads(rng, 2000, 150, 12, 0.2, (10, 30))
    for max_cov in (1, 3, 15):
        chosen = readselection(build(reads, core), max_cov)
        positions = sorted({p for _, variants in reads for p, _, _ in variants})
        index = {p: i for i, p in enumerate(positions)}
        cov = np.zeros(len(positions), dtype=int)
        for i in chosen:
            v = reads[i][1]
            cov[index[v[0][0]]:index[v[-1][0]] + 1] += 1
        assert cov.max() <= max_cov


def test_read_with_one_variant_is_a_value_error():
    rs = core.ReadSet()
    r = core.Read("a", 60, 0, 0)
    r.add_variant(100, 0, 10)
    rs.add(r)
    with pytest.raises(ValueError, match="at least two variants"):
        readselection(rs, 5)


def test_empty_readset():
    assert readselection(core.ReadSet(), 5) == set()


def test_accepts_a_reference_readset():
    ref = reference_core()
    import whatshap.readselect as ref_readselect

    rng = np.random.default_rng(3)
    reads = random_reads(rng, 400, 70, 9, 0.2, (10,))
    rs = build(reads, ref)
    assert readselection(rs, 5) == set(ref_readselect.readselection(rs, 5))


def test_exported_by_the_c_abi():
    assert "whamd_readselection" in _native.EXPORTED_SYMBOLS
