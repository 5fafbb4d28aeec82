"""Inputs of the genotyping tests: small random ReadSets + pedigrees with genotype priors, as flat ProblemArrays, and their
conversion into the reference's own objects."""
import numpy as np

from whatshap_amd import _native


def random_case(seed, n_variants=8, n_reads=10, max_len=5, mode="single", max_coverage=None, recomb=(1, 40), uniform_prior=False,
                phred=(1, 40), blank=0.15):
    """mode: single | trio | quartet.  Reads sorted by first position; positions 10, 20, ..."""
    rng = np.random.default_rng(seed)
    n_ind = {"single": 1, "trio": 3, "quartet": 4}[mode]
    triples = {"single": [], "trio": [(0, 1, 2)], "quartet": [(0, 1, 2), (0, 1, 3)]}[mode]
    reads = []
    cover = np.zeros(n_variants, dtype=int)
    for _ in range(n_reads):
        length = int(rng.integers(2, max_len + 1))
        start = int(rng.integers(0, n_variants - 1))
        idx = [v for v in range(start, min(n_variants, start + length)) if v == start or rng.random() >= blank]
        if len(idx) < 2:
            idx = [start, start + 1] if start + 1 < n_variants else [start - 1, start]
        lo, hi = idx[0], idx[-1]
        if max_coverage is not None and cover[lo:hi + 1].max() >= max_coverage:
            continue
        cover[lo:hi + 1] += 1
        reads.append((int(rng.integers(0, n_ind)), [(10 * (v + 1), int(rng.integers(0, 2)), int(rng.integers(phred[0], phred[1] + 1))) for v in idx]))
    reads.sort(key=lambda r: r[1][0][0])
    read_ptr, pos, alle, qual, samples = [0], [], [], [], []
    for sample, variants in reads:
        samples.append(sample)
        for p, a, q in variants:
            pos.append(p); alle.append(a); qual.append(q)
        read_ptr.append(len(pos))
    positions = np.arange(1, n_variants + 1, dtype=np.uint32) * 10
    if uniform_prior:
        gl = np.full((n_ind, n_variants, 3), 1.0 / 3.0)
    else:
        gl = rng.random((n_ind, n_variants, 3)) + 0.05
        gl /= gl.sum(axis=2, keepdims=True)
    recombcost = rng.integers(recomb[0], recomb[1] + 1, size=n_variants).astype(np.uint32)
    return _native.ProblemArrays(
        np.asarray(read_ptr, dtype=np.uint64), np.asarray(pos, dtype=np.int32), np.asarray(alle, dtype=np.uint8),
        np.asarray(qual, dtype=np.uint32), np.asarray(samples, dtype=np.int32), np.arange(n_ind, dtype=np.uint32),
        np.asarray(triples, dtype=np.uint32).reshape(-1), np.ones((n_ind, n_variants), dtype=np.uint8), gl, recombcost, positions, False,
        n_variants=n_variants)


def reference_likelihoods(problem, ref):
    """[individuals][columns][3] from the REAL whatshap.core.GenotypeDPTable (oracle/_ref/cy)."""
    rs = ref.ReadSet()
    ptr = problem.read_ptr
    for r in range(problem.n_reads):
        read = ref.Read(f"read{r}", 60, 0, int(problem.read_sample_id[r]))
        for i in range(int(ptr[r]), int(ptr[r + 1])):
            read.add_variant(int(problem.var_position[i]), int(problem.var_allele[i]), int(problem.var_quality[i]))
        rs.add(read)
    ids = ref.NumericSampleIds()
    individual_ids = [int(x) for x in problem.individual_id]
    for numeric in range(max(individual_ids) + 1):
        assert ids[str(numeric)] == numeric
    ped = ref.Pedigree(ids)
    n_ind, n_var = problem.n_individuals, problem.n_variants
    gl = problem.genotype_likelihoods.reshape(n_ind, n_var, 3)
    for i, numeric in enumerate(individual_ids):
        gts = [ref.Genotype([0, 1]) for _ in range(n_var)]
        gls = [ref.PhredGenotypeLikelihoods([float(x) for x in gl[i, v]]) for v in range(n_var)]
        ped.add_individual(str(numeric), gts, gls)
    for f, m, c in problem.triple_ids.reshape(-1, 3):
        ped.add_relationship(str(int(f)), str(int(m)), str(int(c)))
    positions = None if problem.positions is None else [int(p) for p in problem.positions]
    table = ref.GenotypeDPTable(ids, rs, [int(x) for x in problem.recombcost], ped, positions)
    n_cols = len(positions) if positions is not None else len(rs.get_positions())
    out = np.zeros((n_ind, n_cols, 3), dtype=np.float64)
    for i, numeric in enumerate(individual_ids):
        for c in range(n_cols):
            likelihoods = table.get_genotype_likelihoods(str(numeric), c)
            out[i, c, :] = [likelihoods[ref.Genotype(g)] for g in ([0, 0], [0, 1], [1, 1])]
    return out
