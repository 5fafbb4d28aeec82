"""Seeded synthetic ReadSets for benchmarks and parity tests.

``synthetic_block`` is the workload of BASELINE.json's configs 2-5: positions 1000*(i+1), two
complementary haplotypes, reads spanning ``step*coverage`` consecutive variants with a new read every
``step`` variants (steady-state coverage == ``coverage``), 2 % allele errors, phred ~ U{5..40}, 10 % of
interior positions dropped (BLANK entries).  ``random_small_instance`` is the tie-heavy generator of
SURVEY.md Appendix A used to pin tie-breaking.

Everything is returned as ``whatshap_amd._native.ProblemArrays`` (the arrays behind the C-ABI views).
"""

from __future__ import annotations

import math
import random
from typing import List, Optional

import numpy as np

from ._native import ProblemArrays


def centimorgen_to_phred(distance: float) -> float:
    """whatshap/pedigree.py:240-250."""
    if distance == 0:
        raise ValueError("Cannot convert genetic distance of zero to phred.")
    if distance < 1e-10:
        return -10.0 * (math.log10(distance) - 2.0)
    p = (1.0 - math.exp(-(2.0 * distance) / 100.0)) / 2.0
    return -10.0 * math.log10(p)


def uniform_recombination_costs(positions, recombrate: float = 1.26) -> np.ndarray:
    """UniformRecombinationCostComputer (whatshap/pedigree.py:104-122)."""
    positions = np.asarray(positions, dtype=np.int64)
    out = np.zeros(positions.size, dtype=np.uint32)
    for i in range(1, positions.size):
        out[i] = round(centimorgen_to_phred((positions[i] - positions[i - 1]) * 1e-6 * recombrate))
    return out


def synthetic_block(n_variants: int, coverage: int, seed: int, step: int = 2, trio: bool = False,
                    distrust_genotypes: bool = False, error_rate: float = 0.02, drop_rate: float = 0.10,
                    n_columns_limit: Optional[int] = None, quartet: bool = False,
                    mixed_genotypes: bool = False, two_trios: bool = False) -> ProblemArrays:
    """One connected phasing block.  ``n_columns_limit`` keeps only the first columns of the SAME
    ReadSet (reads are clipped to the prefix), for bounded CPU-baseline samples.
    ``quartet``: two trios sharing their parents (T = 16), reads round-robin over the four individuals.
    ``two_trios``: two unrelated trios in one table (T = 16, six individuals, four founders: up to 4 cost forms per value).
    ``mixed_genotypes`` (pedigrees): Mendelian-consistent random genotypes instead of all-heterozygous ones."""
    rng = np.random.default_rng(seed)
    n = int(n_variants)
    hap = rng.integers(0, 2, size=n, dtype=np.uint8)  # haplotype 0; haplotype 1 is its complement
    span = step * coverage
    starts = np.arange(0, n, step, dtype=np.int64)
    ends = np.minimum(starts + span, n)
    keep = (ends - starts) >= 2
    starts, ends = starts[keep], ends[keep]
    n_reads = starts.size
    lengths = (ends - starts).astype(np.int64)
    read_of = np.repeat(np.arange(n_reads), lengths)
    offs = np.arange(lengths.sum()) - np.repeat(np.cumsum(lengths) - lengths, lengths)
    var_idx = np.repeat(starts, lengths) + offs
    read_hap = rng.integers(0, 2, size=n_reads, dtype=np.uint8)
    allele = hap[var_idx] ^ read_hap[read_of]
    flip = rng.random(allele.size) < error_rate
    allele = (allele ^ flip.astype(np.uint8)).astype(np.uint8)
    quality = rng.integers(5, 41, size=allele.size, dtype=np.uint32)
    interior = (offs > 0) & (offs < np.repeat(lengths, lengths) - 1)
    dropped = interior & (rng.random(allele.size) < drop_rate)
    if n_columns_limit is not None and n_columns_limit < n:
        dropped |= var_idx >= n_columns_limit
        # the last kept variant of a clipped read must exist in the prefix: un-drop it
        last_in_prefix = (var_idx == n_columns_limit - 1)
        dropped &= ~last_in_prefix
    keepv = ~dropped
    read_of, var_idx, allele, quality = read_of[keepv], var_idx[keepv], allele[keepv], quality[keepv]
    counts = np.bincount(read_of, minlength=n_reads)
    ok_reads = counts >= 2
    if not ok_reads.all():
        sel = ok_reads[read_of]
        read_of, var_idx, allele, quality = read_of[sel], var_idx[sel], allele[sel], quality[sel]
        remap = np.cumsum(ok_reads) - 1
        read_of = remap[read_of]
        counts = counts[ok_reads]
        n_reads = int(ok_reads.sum())
    read_ptr = np.zeros(n_reads + 1, dtype=np.uint64)
    read_ptr[1:] = np.cumsum(counts)
    n_cols = n if n_columns_limit is None else min(n, n_columns_limit)
    positions = (1000 * (np.arange(n_cols, dtype=np.int64) + 1)).astype(np.uint32)
    var_position = (1000 * (var_idx + 1)).astype(np.int32)
    if trio or quartet or two_trios:
        n_ind = 6 if two_trios else (4 if quartet else 3)
        sample = (np.arange(n_reads) % n_ind).astype(np.int32)  # father, mother, child (, second child) round-robin
        individual_id = np.arange(n_ind, dtype=np.uint32)
        triples = np.array([0, 1, 2, 3, 4, 5] if two_trios else ([0, 1, 2, 0, 1, 3] if quartet else [0, 1, 2]), dtype=np.uint32)
        genotype = np.ones((n_ind, n_cols), dtype=np.uint8)
        if mixed_genotypes:
            grng = np.random.default_rng(seed + 7919)
            cols = np.arange(n_cols)
            for fam in range(0, n_ind, 3) if two_trios else (0,):
                fa = grng.integers(0, 2, size=(2, n_cols))
                mo = grng.integers(0, 2, size=(2, n_cols))
                genotype[fam] = fa.sum(axis=0)
                genotype[fam + 1] = mo.sum(axis=0)
                for child in range(fam + 2, fam + 3 if two_trios else n_ind):
                    genotype[child] = fa[grng.integers(0, 2, size=n_cols), cols] + mo[grng.integers(0, 2, size=n_cols), cols]
        recomb = np.zeros(n_cols, dtype=np.uint32)
        if n_cols > 1:
            recomb[1:] = round(centimorgen_to_phred(1000 * 1e-6 * 1.26))
        gl = np.tile(np.array([30.0, 0.0, 30.0]), (n_ind, n_cols, 1)) if distrust_genotypes else None
    else:
        sample = np.zeros(n_reads, dtype=np.int32)
        individual_id = np.array([0], dtype=np.uint32)
        triples = np.zeros(0, dtype=np.uint32)
        genotype = np.ones((1, n_cols), dtype=np.uint8)
        recomb = np.ones(n_cols, dtype=np.uint32)
        gl = np.tile(np.array([30.0, 0.0, 30.0]), (1, n_cols, 1)) if distrust_genotypes else None
    return ProblemArrays(read_ptr, var_position, allele, quality, sample, individual_id, triples, genotype, gl,
                         recomb, positions, distrust_genotypes, n_variants=n_cols)


def irregular_block(n_variants: int, coverage: int, seed: int, mean_length: Optional[float] = None,
                    error_rate: float = 0.02, drop_rate: float = 0.10) -> ProblemArrays:
    """A single-individual block whose read layout is NOT the planner's best case: reads start at random (Poisson
    number of new reads per variant), lengths are geometric (mean ``mean_length`` variants, default 0.8 * coverage, at
    least 2), and a read that would push the physical coverage of any column it spans beyond ``coverage`` is not
    generated (what read selection guarantees, whatshap/readselect.pyx:140-145).  Same allele / quality / BLANK model as
    ``synthetic_block``.  bench.py's irregular-layout line uses seed 7."""
    rng = np.random.default_rng(seed)
    n = int(n_variants)
    mean_len = float(mean_length) if mean_length else 0.8 * coverage
    lam = 1.15 * coverage / mean_len          # a little more than the cap admits: the cap binds most of the time
    n_new = rng.poisson(lam, size=n)
    cov = np.zeros(n + 1, dtype=np.int32)      # difference array of the physical coverage
    active_end: List[int] = []
    starts, ends = [], []
    p_stop = 1.0 / max(mean_len - 1.0, 1.0)
    running = 0
    import heapq
    for i in range(n - 1):
        while active_end and active_end[0] <= i:
            heapq.heappop(active_end)
        running = len(active_end)
        for _ in range(int(n_new[i])):
            if running >= coverage:
                break
            length = 2 + int(rng.geometric(p_stop)) - 1
            e = min(i + length, n)
            if e - i < 2:
                continue
            starts.append(i)
            ends.append(e)
            heapq.heappush(active_end, e)
            running += 1
    starts = np.asarray(starts, dtype=np.int64)
    ends = np.asarray(ends, dtype=np.int64)
    order = np.argsort(starts, kind="stable")
    starts, ends = starts[order], ends[order]
    n_reads = starts.size
    hap = rng.integers(0, 2, size=n, dtype=np.uint8)
    lengths = ends - starts
    read_of = np.repeat(np.arange(n_reads), lengths)
    offs = np.arange(lengths.sum()) - np.repeat(np.cumsum(lengths) - lengths, lengths)
    var_idx = np.repeat(starts, lengths) + offs
    read_hap = rng.integers(0, 2, size=n_reads, dtype=np.uint8)
    allele = hap[var_idx] ^ read_hap[read_of]
    allele = (allele ^ (rng.random(allele.size) < error_rate).astype(np.uint8)).astype(np.uint8)
    quality = rng.integers(5, 41, size=allele.size, dtype=np.uint32)
    interior = (offs > 0) & (offs < np.repeat(lengths, lengths) - 1)
    keepv = ~(interior & (rng.random(allele.size) < drop_rate))
    read_of, var_idx, allele, quality = read_of[keepv], var_idx[keepv], allele[keepv], quality[keepv]
    counts = np.bincount(read_of, minlength=n_reads)
    read_ptr = np.zeros(n_reads + 1, dtype=np.uint64)
    read_ptr[1:] = np.cumsum(counts)
    positions = (1000 * (np.arange(n, dtype=np.int64) + 1)).astype(np.uint32)
    return ProblemArrays(read_ptr, (1000 * (var_idx + 1)).astype(np.int32), allele, quality, np.zeros(n_reads, dtype=np.int32),
                         np.array([0], dtype=np.uint32), np.zeros(0, dtype=np.uint32), np.ones((1, n), dtype=np.uint8), None,
                         np.ones(n, dtype=np.uint32), positions, False, n_variants=n)


def clip_to_columns(p: ProblemArrays, n_columns: int) -> ProblemArrays:
    """The first ``n_columns`` columns of a block: reads are clipped to the prefix, reads left with fewer than two
    variants are dropped (bounded CPU-baseline samples of a workload without its own prefix option)."""
    n_columns = int(min(n_columns, p.n_variants))
    limit = int(p.positions[n_columns - 1])
    lengths = np.diff(p.read_ptr).astype(np.int64)
    read_of = np.repeat(np.arange(p.n_reads), lengths)
    keep = p.var_position <= limit
    counts = np.bincount(read_of[keep], minlength=p.n_reads)
    ok = counts >= 2
    keep &= ok[read_of]
    read_ptr = np.zeros(int(ok.sum()) + 1, dtype=np.uint64)
    read_ptr[1:] = np.cumsum(counts[ok])
    n_ind = p.individual_id.size
    gl = None if p.genotype_likelihoods is None else p.genotype_likelihoods.reshape(n_ind, p.n_variants, 3)[:, :n_columns]
    return ProblemArrays(read_ptr, p.var_position[keep], p.var_allele[keep], p.var_quality[keep], p.read_sample_id[ok], p.individual_id,
                         p.triple_ids, p.genotype.reshape(n_ind, p.n_variants)[:, :n_columns], gl, p.recombcost[:n_columns],
                         p.positions[:n_columns], p.distrust_genotypes, n_variants=n_columns)


def random_small_instance(rng: random.Random, mode: Optional[str] = None, max_variants: int = 9,
                          max_reads: int = 7, allow_conflict: bool = True) -> ProblemArrays:
    """Tie-heavy random instance: few phred values, gapped reads, random
    recombination costs, optional extra positions, optional genotype inconsistency."""
    if mode is None:
        mode = rng.choice(["single", "trio", "quartet"])
    n_var = rng.randint(2, max_variants)
    n_reads = rng.randint(1, max_reads)
    n_ind = {"single": 1, "trio": 3, "quartet": 4, "three_children": 5, "three_trios": 9, "big_family": 7}[mode]
    var_positions = [10 * (i + 1) for i in range(n_var)]
    reads = []
    for _ in range(n_reads):
        s = rng.randint(0, n_var - 2)
        e = rng.randint(s + 1, n_var - 1)
        variants = []
        for i in range(s, e + 1):
            if i in (s, e) or rng.random() < 0.75:
                variants.append((var_positions[i], rng.randint(0, 1), rng.choice([1, 1, 2, 7])))
        reads.append((variants, rng.randrange(n_ind)))
    reads.sort(key=lambda r: r[0][0][0])  # by first position; ties keep generation order (any order is a valid sorted ReadSet)
    positions = list(var_positions)
    if rng.random() < 0.6:
        extra = set()
        for _ in range(rng.randint(1, 3)):
            extra.add(rng.choice([5, 15, 25, 35, 45, 10 * n_var + 5, 10 * n_var + 15]))
        positions = sorted(set(positions) | extra)
    n_cols = len(positions)
    distrust = rng.random() < 0.5
    genotype = np.ones((n_ind, n_cols), dtype=np.uint8)
    triples: List[int] = []
    if mode == "trio":
        triples = [0, 1, 2]
    elif mode == "quartet":
        triples = [0, 1, 2, 0, 1, 3]
    elif mode == "three_children":   # T = 64: two parents, three children
        triples = [0, 1, 2, 0, 1, 3, 0, 1, 4]
    elif mode == "three_trios":      # T = 64: three unrelated trios in one table
        triples = [0, 1, 2, 3, 4, 5, 6, 7, 8]
    elif mode == "big_family":       # T = 16, seven individuals: grandparents (0, 1) -> father 2; mother 3; child 4; two unrelated
        triples = [0, 1, 2, 2, 3, 4]
    if mode != "single" and rng.random() < 0.5:
        for c in range(n_cols):
            hap = {}
            for ind in range(n_ind):
                trio = [tr for tr in range(len(triples) // 3) if triples[3 * tr + 2] == ind]
                if trio:   # a child: one haplotype of each parent (parents come first in every mode)
                    fa, mo = triples[3 * trio[0]], triples[3 * trio[0] + 1]
                    hap[ind] = [rng.choice(hap[fa]), rng.choice(hap[mo])]
                else:
                    hap[ind] = [rng.randint(0, 1), rng.randint(0, 1)]
                genotype[ind, c] = sum(hap[ind])
    if allow_conflict and rng.random() < 0.3:
        genotype[rng.randrange(n_ind), rng.randrange(n_cols)] = rng.randint(0, 2)
    gl = np.zeros((n_ind, n_cols, 3), dtype=np.float64)
    for i in range(n_ind):
        for c in range(n_cols):
            for g in range(3):
                gl[i, c, g] = rng.choice([0, 0, 3, 10])
    recomb = np.array([rng.choice([0, 1, 2, 5]) for _ in range(n_cols)], dtype=np.uint32)
    read_ptr = [0]
    pos, alle, qual, samples = [], [], [], []
    for variants, sample in reads:
        for p, a, q in variants:
            pos.append(p)
            alle.append(a)
            qual.append(q)
        read_ptr.append(len(pos))
        samples.append(sample)
    return ProblemArrays(read_ptr, pos, alle, qual, samples, np.arange(n_ind, dtype=np.uint32),
                         np.asarray(triples, dtype=np.uint32), genotype, gl, recomb,
                         np.asarray(positions, dtype=np.uint32), distrust, n_variants=n_cols)
