"""Soak of the round-4 / round-5 paths on the device against the oracle, beyond the test-suite: random batches of irregular / regular / tie-heavy tables
(single individuals with mixed genotypes -- Y-form and general runs in one table --, components, trios with trusted and untrusted genotypes,
quartets) solved through whamd_dptable_enqueue_many (shared launches), with and without the shared_launches layout (eight cells per thread for
the wide ones), and alone.  Prints the number of mismatches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from helpers import table_solution, first_difference
from whatshap_amd import _native
from whatshap_amd.synthetic import irregular_block, synthetic_block
from gpu_multiblock import chromosome


def variant(p, rng, ties, mixed):
    q = p.var_quality
    if ties:
        q = rng.choice(np.array([3, 6], dtype=np.uint32), size=q.size)
    g = p.genotype.reshape(p.n_individuals, -1)
    if mixed and p.n_individuals == 1:   # homozygous columns: runs that cannot take the Y form, conversions at their boundaries
        g = rng.choice(np.array([0, 1, 1, 1, 1, 2], dtype=np.uint8), size=g.shape)
    return _native.ProblemArrays(p.read_ptr, p.var_position, p.var_allele, q, p.read_sample_id, p.individual_id, p.triple_ids, g, p.genotype_likelihoods,
                                 p.recombcost, p.positions, p.distrust_genotypes, n_variants=p.n_variants)


def make(seed):
    rng = np.random.default_rng(seed)
    kind = seed % 8
    if kind == 0:
        p = irregular_block(int(rng.integers(150, 500)), int(rng.integers(16, 21)), seed=seed, mean_length=float(rng.integers(4, 14)))
    elif kind == 1:
        p = synthetic_block(int(rng.integers(200, 700)), int(rng.integers(17, 21)), seed=seed, step=int(rng.integers(1, 4)))
    elif kind == 2:
        p = irregular_block(int(rng.integers(200, 900)), int(rng.integers(9, 16)), seed=seed)
    elif kind == 3:
        p = chromosome(int(rng.integers(3, 9)), int(rng.integers(8, 14)), seed=seed, max_len=140)
    elif kind == 4:
        p = synthetic_block(int(rng.integers(150, 400)), int(rng.integers(8, 13)), seed=seed, trio=True, distrust_genotypes=bool(rng.integers(0, 2)))
    elif kind == 5:
        p = synthetic_block(int(rng.integers(150, 300)), int(rng.integers(7, 11)), seed=seed, quartet=True, distrust_genotypes=bool(rng.integers(0, 2)))   # (round 5: factorised lines of a quartet)
    elif kind == 6:
        p = synthetic_block(int(rng.integers(300, 1500)), int(rng.integers(12, 19)), seed=seed, distrust_genotypes=bool(rng.integers(0, 2)))
    else:
        p = synthetic_block(int(rng.integers(200, 500)), int(rng.integers(18, 21)), seed=seed, error_rate=0.1, drop_rate=0.3)
    return variant(p, rng, ties=bool(rng.integers(0, 2)), mixed=bool(rng.integers(0, 3) == 0))


t0 = time.time()
bad = n = 0
n_batches = int(os.environ.get("WHAMD_SOAK_BATCHES", "12"))
for b in range(n_batches):
    problems = [make(10000 + 100 * b + i) for i in range(10 + b % 7)]
    want = [table_solution(oracle.OracleTable(p)) for p in problems]
    for opts in (None, {"shared_launches": "1"}, {"slot_r": "3", "slot_l": "12"}):
        tables = [_native.NativeTable(p, solve=False, options=opts) for p in problems]
        _native.enqueue_many(tables)
        _native.wait_many(tables)
        for i, (t, w) in enumerate(zip(tables, want)):
            n += 1
            got = table_solution(t)
            if got != w:
                bad += 1
                print("MISMATCH batch", b, "table", i, opts, first_difference(w, got), flush=True)
            t.close()
    for i, (p, w) in enumerate(zip(problems, want)):   # alone with the default layout (round 5: X runs on the thread's own LDS lines, narrow tables packed onto one XCD)
        n += 1
        t = _native.NativeTable(p)
        if table_solution(t) != w:
            bad += 1
            print("MISMATCH alone", b, i, first_difference(w, table_solution(t)), flush=True)
        t.close()
    for i, (p, w) in enumerate(zip(problems, want)):   # alone, eight cells per thread
        n += 1
        t = _native.NativeTable(p, options={"slot_r": "3", "slot_l": "12"})
        if table_solution(t) != w:
            bad += 1
            print("MISMATCH alone slot_r=3", b, i, first_difference(w, table_solution(t)), flush=True)
        t.close()
    print(f"batch {b}: {len(problems)} tables, {n} solves so far, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("group soak:", n, "solves,", bad, "mismatches")
sys.exit(1 if bad else 0)
