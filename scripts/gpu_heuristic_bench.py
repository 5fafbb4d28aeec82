#!/usr/bin/env python3
"""Measurement of the PedMecHeuristic row: the persistent single-workgroup kernel on one MI355X next to the
compiled reference's solve() on one host core, on synthetic ReadSets at coverages the exact DP cannot afford.

    python scripts/gpu_heuristic_bench.py [--variants 20000] [--coverage 30] [--row-limit 256] [--trio]

One JSON line: variant-columns/s of the device solve (HIP events around the kernel) and of the whole call from host arrays,
the reference's rate on the same input (full length), and whether every output is identical."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from whatshap_amd import _native  # noqa: E402
from whatshap_amd.synthetic import synthetic_block  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=20000)
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--row-limit", type=int, default=256)
    ap.add_argument("--trio", action="store_true")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-reference", action="store_true")
    args = ap.parse_args()
    p = synthetic_block(n_variants=args.variants, coverage=args.coverage, seed=args.seed, trio=args.trio)
    _native.pedmec_heuristic(p, row_limit=args.row_limit)   # warm-up
    dev, wall, got = [], [], None
    for _ in range(args.steps):
        t0 = time.perf_counter()
        got = _native.pedmec_heuristic(p, row_limit=args.row_limit)
        wall.append(time.perf_counter() - t0)
        dev.append(got["stats"]["device_ms"] / 1e3)
    dev_s, wall_s = sorted(dev)[len(dev) // 2], sorted(wall)[len(wall) // 2]
    out = {"metric": f"variant-columns/s of PedMecHeuristic.solve at max-coverage {args.coverage}, row limit {args.row_limit}", "value": args.variants / dev_s,
           "unit": "variant-columns/s", "end_to_end": args.variants / wall_s, "device_ms": dev_s * 1e3, "wall_ms": wall_s * 1e3,
           "config": {"workload": f"synthetic {'trio' if args.trio else 'single individual'}, {args.variants} SNVs, max-coverage {args.coverage}", "row_limit": args.row_limit},
           "stats": got["stats"], "dtype": "f32 (scores) + bit sets", "steps": args.steps}
    if not args.no_reference:
        import oracle
        from heuristic_cases import result_tuple

        ref = oracle.ReferenceHeuristic(p, row_limit=args.row_limit)
        out["cpu_baseline"] = {"value": args.variants / ref.solve_seconds(), "unit": "variant-columns/s", "cores": 1, "kind": "reference",
                               "sample": f"PedMecHeuristic::solve() of the compiled reference on the same ReadSet, full length, {ref.solve_seconds():.2f} s"}
        out["identical_to_reference"] = result_tuple(got) == oracle.heuristic_tuple(ref)
        out["speedup_vs_cpu_baseline_device_only"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
