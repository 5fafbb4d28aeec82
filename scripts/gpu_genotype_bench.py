#!/usr/bin/env python3
"""Measurement of the genotyping path: GenotypeDPTable on one MI355X next to the REAL reference
class timed on one host core (oracle/_ref/cy, a bounded prefix of the same ReadSet), on the synthetic ReadSets of the
phasing benchmark with uniform genotype priors (what `whatshap genotype` passes when no priors are given).

    python scripts/gpu_genotype_bench.py [--variants 50000] [--coverage 15] [--trio] [--steps 3] [--cpu-columns 60]

Prints one JSON line: variant-columns/s (constructor = the whole forward-backward pass, inputs as host arrays; HIP-event
device time reported alongside), cells/s, parity of the timed prefix against the reference (max abs difference)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from whatshap_amd import _native  # noqa: E402
from whatshap_amd.synthetic import synthetic_block  # noqa: E402


def with_priors(p, seed=None):
    n_ind, n_var = p.n_individuals, p.n_variants
    if seed is None:
        gl = np.full((n_ind, n_var, 3), 1.0 / 3.0)
    else:
        gl = np.random.default_rng(seed).random((n_ind, n_var, 3)) + 0.05
        gl /= gl.sum(axis=2, keepdims=True)
    return _native.ProblemArrays(p.read_ptr, p.var_position, p.var_allele, p.var_quality, p.read_sample_id, p.individual_id, p.triple_ids,
                                 p.genotype.reshape(n_ind, n_var), gl, p.recombcost, p.positions, False, n_variants=n_var)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=50000)
    ap.add_argument("--coverage", type=int, default=15)
    ap.add_argument("--trio", action="store_true")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--cpu-columns", type=int, default=60, help="columns of the prefix the reference is timed on (0: skip)")
    args = ap.parse_args()
    problem = with_priors(synthetic_block(n_variants=args.variants, coverage=args.coverage, seed=args.seed, trio=args.trio))
    n = int(problem.positions.size)
    for _ in range(args.warmup):
        _native.genotype_likelihoods(problem, n, window=args.window)
    wall, dev = [], []
    stats = None
    for _ in range(args.steps):
        t0 = time.perf_counter()
        gl, stats = _native.genotype_likelihoods(problem, n, window=args.window)
        wall.append(time.perf_counter() - t0)
        dev.append(stats["total_ms"] / 1e3)
    out = {
        "metric": "variant-columns/s of GenotypeDPTable (constructor + all likelihoods), max-coverage %d" % args.coverage,
        "value": n / float(np.median(wall)), "unit": "variant-columns/s", "value_note": "median over the steps (ms_steps lists them: the first call of a process pays for the device heap growing by tens of GB)",
        "device_only_value": n / float(np.median(dev)), "cells_per_s": stats["n_cells"] / float(np.median(dev)),
        "ms_per_step": 1e3 * float(np.median(wall)), "ms_steps": [round(1e3 * w, 1) for w in wall], "device_ms_per_step": 1e3 * float(np.median(dev)), "steps": args.steps, "warmup": args.warmup,
        "dtype": "f64", "data": "synthetic", "config": {"workload": "synthetic %s, %d SNVs, max-coverage %d, uniform genotype priors" % (
            "trio" if args.trio else "single individual", args.variants, args.coverage), "window": stats["window"], "launches": stats["launches"],
            "transmission_values": stats["transmissions"]},
        "stats": stats,
    }
    if args.cpu_columns:
        from genotype_cases import reference_likelihoods
        from oracle import build_cython_ref

        if build_cython_ref.available():
            ref = build_cython_ref.import_reference()
            ramp = 2 * args.coverage
            times = {}
            for cols in (ramp, ramp + args.cpu_columns):   # steady-state columns = difference of two prefixes
                prefix = with_priors(synthetic_block(n_variants=args.variants, coverage=args.coverage, seed=args.seed, trio=args.trio, n_columns_limit=cols))
                t0 = time.perf_counter()
                want = reference_likelihoods(prefix, ref)
                times[cols] = time.perf_counter() - t0
            got, _ = _native.genotype_likelihoods(prefix, int(prefix.positions.size))
            steady = max(times[ramp + args.cpu_columns] - times[ramp], 1e-9)
            out["cpu_baseline"] = {"value": args.cpu_columns / steady, "unit": "variant-columns/s", "cores": 1, "kind": "reference",
                                   "sample": "%d steady-state columns (prefix of %d minus prefix of %d columns of the same ReadSet), whatshap.core.GenotypeDPTable constructor + every get_genotype_likelihoods, %.1f s" % (
                                       args.cpu_columns, ramp + args.cpu_columns, ramp, times[ramp + args.cpu_columns])}
            out["parity_prefix_max_abs_diff"] = float(np.abs(got - want).max())
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
