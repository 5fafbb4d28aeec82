"""Inputs of the PedMecHeuristic tests: random tie-heavy instances and synthetic blocks in the shape the
reference's solver accepts -- sorted reads, positive recombination costs (a zero cost makes the reference index an empty
vector in getOptPhasing, src/pedmecheuristic.cpp:493-497,523), every individual with at least one read (the reference looks
genotypes up by the RANK of the sample id among the ids that occur, :49-82)."""
import random

import numpy as np

from whatshap_amd import _native
from whatshap_amd.synthetic import random_small_instance, synthetic_block


def positive_recomb(p):
    n_ind = p.n_individuals
    return _native.ProblemArrays(p.read_ptr, p.var_position, p.var_allele, p.var_quality, p.read_sample_id, p.individual_id, p.triple_ids,
                                 p.genotype.reshape(n_ind, -1), None, np.maximum(p.recombcost, 1), p.positions, p.distrust_genotypes,
                                 n_variants=p.n_variants)


def random_cases(seed, count, modes=("single", "trio", "quartet")):
    """(name, problem, row_limit) of `count` random instances per mode whose individuals all have reads."""
    rng = random.Random(seed)
    out = []
    for mode in modes:
        kept = 0
        while kept < count:
            p = positive_recomb(random_small_instance(rng, mode=mode, allow_conflict=False))
            row_limit = rng.choice([2, 4, 8, 256])
            if len(set(p.read_sample_id.tolist())) != p.n_individuals:
                continue
            out.append((f"{mode}_{kept}", p, row_limit))
            kept += 1
    return out


SYNTHETIC = [
    (dict(n_variants=300, coverage=12, seed=3, trio=True), 64),
    (dict(n_variants=400, coverage=20, seed=4), 32),
    (dict(n_variants=300, coverage=10, seed=5, quartet=True), 64),
    (dict(n_variants=200, coverage=12, seed=6, trio=True, distrust_genotypes=True), 32),
    (dict(n_variants=300, coverage=16, seed=7, trio=True, error_rate=0.1), 256),
    (dict(n_variants=250, coverage=14, seed=8, trio=True, mixed_genotypes=True), 128),
    (dict(n_variants=500, coverage=26, seed=9, error_rate=0.05), 256),          # beyond the exact DP's 25 reads per column
    (dict(n_variants=160, coverage=70, seed=10, error_rate=0.05), 32),          # three words per bipartition
]


def synthetic_cases():
    return [("synthetic_" + "_".join(f"{k}{v}" for k, v in kw.items()) + f"_rows{rl}", synthetic_block(**kw), rl) for kw, rl in SYNTHETIC]


def result_tuple(out):
    """whatshap_amd._native.pedmec_heuristic(...) in the comparable form of oracle.heuristic_tuple."""
    return {"score": out["score"], "bipartition": out["bipartition"].tolist(), "transmission": out["transmission"].tolist(),
            "haplotypes": out["haplotypes"].tolist(), "mutated": out["mutated"].tolist()}
