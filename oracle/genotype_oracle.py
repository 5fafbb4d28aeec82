"""TEST INFRASTRUCTURE -- CPU restatement of the reference's GenotypeDPTable.

Plain Python / numpy loops over every bipartition, in ``numpy.longdouble`` like the reference's ``long double``; for small
inputs only (tests, smoke).  Never imported by the product (``whatshap_amd/``): the product path is the HIP library.

Follows, line by line where it matters:
  columns / indexing      src/columniterator.cpp:91-139, src/columnindexingscheme.cpp:7-34,62-85 (via whatshap_amd's flat view)
  partitions              src/pedigreepartitions.cpp:7-42
  emission                src/genotypecolumncostcomputer.cpp:26-103 (phred 0 -> 0.9999; bit 0 = "entry_in_partition1")
  transitions / priors    src/transitionprobabilitycomputer.cpp:22-90
  backward pass           src/genotypedptable.cpp:200-289
  forward pass, output    src/genotypedptable.cpp:292-441, 444-451
The scaling of the reference (scaling_parameters) is a per-column constant that cancels in the normalised output; this
restatement normalises every column to sum 1 instead.  Pinned against the compiled reference in tests/test_genotype_oracle.py.
"""
import numpy as np

LD = np.longdouble


def _columns(problem):
    """Per column: list of (read index, allele 0/1/2, phred, individual index) of the active reads, in read order."""
    ptr = problem.read_ptr
    pos = problem.var_position
    n_reads = problem.n_reads
    positions = sorted(set(int(x) for x in pos)) if problem.positions is None else [int(x) for x in problem.positions]
    index = {p: i for i, p in enumerate(positions)}
    ind_of = {int(v): i for i, v in enumerate(problem.individual_id)}   # later duplicates win (Pedigree::id_to_index)
    columns = [[] for _ in positions]
    for r in range(n_reads):
        lo, hi = int(ptr[r]), int(ptr[r + 1])
        first, last = index[int(pos[lo])], index[int(pos[hi - 1])]
        have = {int(pos[i]): i for i in range(lo, hi)}
        for c in range(first, last + 1):
            i = have.get(positions[c])
            if i is None:
                columns[c].append((r, 2, 0, ind_of[int(problem.read_sample_id[r])]))
            else:
                columns[c].append((r, int(problem.var_allele[i]), int(problem.var_quality[i]), ind_of[int(problem.read_sample_id[r])]))
    return positions, columns


def _phred_probability(q):
    return LD("0.9999") if q == 0 else LD(10) ** (-LD(q) / LD(10))


def genotype_likelihoods(problem, h2p_tables=None):
    """[individuals][columns][3] normalised genotype likelihoods as numpy.longdouble."""
    n_ind = problem.n_individuals
    triples_ids = problem.triple_ids.reshape(-1, 3)
    ind_of = {int(v): i for i, v in enumerate(problem.individual_id)}
    triples = [(ind_of[int(f)], ind_of[int(m)], ind_of[int(c)]) for f, m, c in triples_ids]
    T = 4 ** len(triples)
    positions, columns = _columns(problem)
    n = len(positions)
    out = np.zeros((n_ind, n, 3), dtype=LD)
    if n == 0:
        return out
    h2p, P = h2p_tables if h2p_tables is not None else partitions_for(problem)
    A = 1 << P
    gl_prior = problem.genotype_likelihoods.reshape(n_ind, problem.n_variants, 3)
    recomb = list(problem.recombcost) + [problem.recombcost[-1]] * max(0, n - len(problem.recombcost)) if len(problem.recombcost) else [0] * n

    def transition(c):
        r = LD(10) ** (-LD(int(recomb[c])) / LD(10))
        nt = len(triples)
        bern = [r ** LD(x) * (LD(1) - r) ** LD(2 * nt - x) for x in range(2 * nt + 1)]
        norm = sum(bern[bin(j).count("1")] for j in range(T))
        return [[bern[bin(i ^ j).count("1")] / norm for j in range(T)] for i in range(T)]

    def prior(c):
        table = np.zeros((T, A), dtype=LD)
        for i in range(T):
            keys, count = [], {}
            for a in range(A):
                pr = LD(1)
                key = []
                for s in range(n_ind):
                    g = ((a >> h2p[i][s][0]) & 1) + ((a >> h2p[i][s][1]) & 1)
                    pr *= LD(gl_prior[s, c, g])
                    key.append(g)
                key = tuple(key)
                table[i, a] = pr
                keys.append(key)
                count[key] = count.get(key, 0) + 1
            for a in range(A):
                table[i, a] /= LD(count[keys[a]])
            table[i] /= table[i].sum()
        return table

    def cell_costs(col, x):
        """[T][A] emission of bipartition x."""
        res = np.ones((T, A), dtype=LD)
        for i in range(T):
            W = np.ones((P, 2), dtype=LD)
            for j, (_r, allele, q, ind) in enumerate(col):
                if allele == 2:
                    continue
                bit = (x >> j) & 1
                part = h2p[i][ind][1 if bit == 0 else 0]
                pe = _phred_probability(q)
                W[part][allele] *= LD(1) - pe
                W[part][1 - allele] *= pe
            for a in range(A):
                v = LD(1)
                for p in range(P):
                    v *= W[p][(a >> p) & 1]
                res[i, a] = v
        return res

    k = [len(col) for col in columns]
    ids = [[e[0] for e in col] for col in columns]
    # forward mask of column c: reads that are also active in column c + 1; they are the low bits there
    fwd_bits = []
    for c in range(n):
        nxt = set(ids[c + 1]) if c + 1 < n else set()
        fwd_bits.append([j for j, r in enumerate(ids[c]) if r in nxt])
    b = [0] + [len(fwd_bits[c - 1]) for c in range(1, n)]

    def fwd_index(c, x):
        y = 0
        for o, j in enumerate(fwd_bits[c]):
            y |= ((x >> j) & 1) << o
        return y

    trans = [transition(c) for c in range(n)]
    priors = [prior(c) for c in range(n)]
    costs = [[cell_costs(columns[c], x) for x in range(1 << k[c])] for c in range(n)]
    # backward: B[c][y][i] indexed by the forward projection of column c
    B = [None] * n
    for c in range(n - 1, 0, -1):
        cur = np.zeros((1 << b[c], T), dtype=LD)
        for x in range(1 << k[c]):
            y = x & ((1 << b[c]) - 1)
            for i in range(T):
                beta = LD(1) if c == n - 1 else B[c][fwd_index(c, x), i]
                s = (priors[c][i] * costs[c][x][i]).sum() * beta
                for j in range(T):
                    cur[y, j] += s * trans[c][j][i]
        B[c - 1] = cur / cur.sum()
    prev = None
    for c in range(n):
        cur = np.zeros((1 << len(fwd_bits[c]), T), dtype=LD)
        like = np.zeros((n_ind, 3), dtype=LD)
        norm = LD(0)
        for x in range(1 << k[c]):
            yb = x & ((1 << b[c]) - 1)
            yf = fwd_index(c, x)
            for i in range(T):
                sp = LD(1) if c == 0 else sum(prev[yb, j] * trans[c][j][i] for j in range(T))
                beta = LD(1) if c == n - 1 else B[c][yf, i]
                for a in range(A):
                    fw = sp * costs[c][x][i, a] * priors[c][i, a]
                    fb = fw * beta
                    norm += fb
                    cur[yf, i] += fw
                    for s in range(n_ind):
                        g = ((a >> h2p[i][s][0]) & 1) + ((a >> h2p[i][s][1]) & 1)
                        like[s, g] += fb
        out[:, c, :] = like / norm
        prev = cur / cur.sum()
    return out


def partitions_for(problem):
    """(h2p[T][individual][2], P) of a problem, following PedigreePartitions (src/pedigreepartitions.cpp:7-42): founders get
    partitions (2r, 2r + 1) in individual order; a child's haplotype 0 is its father's haplotype !bit(t, 2 * trio), its
    haplotype 1 its mother's haplotype !bit(t, 2 * trio + 1) (:38-41)."""
    n_ind = problem.n_individuals
    ind_of = {int(v): i for i, v in enumerate(problem.individual_id)}
    triples = [(ind_of[int(f)], ind_of[int(m)], ind_of[int(c)]) for f, m, c in problem.triple_ids.reshape(-1, 3)]
    T = 4 ** len(triples)
    tables = []
    P = 0
    for t in range(T):
        h2p = [[-1, -1] for _ in range(n_ind)]
        children = {c for _, _, c in triples}
        p = 0
        for i in range(n_ind):
            if i not in children:
                h2p[i] = [p, p + 1]
                p += 2
        P = p
        pending = True
        while pending:
            pending = False
            for trio, (f, m, c) in enumerate(triples):
                if h2p[c][0] >= 0:
                    continue
                if h2p[f][0] < 0 or h2p[m][0] < 0:
                    pending = True
                    continue
                h2p[c][0] = h2p[f][0 if (t >> (2 * trio)) & 1 else 1]
                h2p[c][1] = h2p[m][0 if (t >> (2 * trio + 1)) & 1 else 1]
        tables.append(h2p)
    return tables, P
