#!/usr/bin/env python3
"""Benchmark of the wMEC/PedMEC hot path (BASELINE.json metric) on 1..N MI355X.

A *step* is one complete pass of the hot path -- forward DP over every column + backtrace to the index path --
over the rank's synthetic phasing block(s), with the flattened input already resident in HBM.
At N=1 the workload is BASELINE.json configs[2]: synthetic diploid single-individual ReadSet, 200 000 het SNVs,
max-coverage 20 (2^20 bipartitions per column).  For N>1 every rank solves its own block of the same shape
(seed 3 + rank): independent blocks, no data-path collective (weak scaling), timing = max over ranks.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--variants", type=int, default=200000, help="columns per block (BASELINE configs[2]: 200000)")
    ap.add_argument("--coverage", type=int, default=20)
    ap.add_argument("--blocks-per-gpu", type=int, default=1)
    ap.add_argument("--trio", action="store_true", help="configs[3]-shaped workload (trio PedMEC) instead")
    ap.add_argument("--path", default="auto")
    ap.add_argument("--cpu-baseline-columns", type=int, default=-1,
                    help="columns of the same ReadSet timed on the compiled reference (default: ~15-20 s worth; 0 = skip)")
    return ap.parse_args()


def cpu_baseline(args, n_columns):
    """The compiled reference (oracle/_ref, single thread) on a bounded prefix of the same workload."""
    import oracle
    from whatshap_amd.synthetic import synthetic_block

    kind = "reference" if oracle.have_reference() else "port"
    table_cls = oracle.ReferenceTable if kind == "reference" else oracle.OracleTable
    problem = synthetic_block(args.variants, args.coverage, seed=3, trio=args.trio, n_columns_limit=n_columns)
    t0 = time.perf_counter()
    table = table_cls(problem)
    score = table.optimal_score()
    table.super_reads()
    table.partitioning()
    dt = time.perf_counter() - t0
    cols = table.n_columns
    return {
        "value": cols / dt,
        "unit": "variant-columns/s",
        "cores": 1,
        "kind": kind,
        "sample": f"first {cols} columns of the same seeded ReadSet (coverage {args.coverage}"
                  f"{', trio' if args.trio else ''}), constructor + 3 getters, {dt:.1f} s wall, optimal cost {score}",
        "seconds": dt,
    }


def measured_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), valid for the
    default workload only; bench.py cannot run the profiler itself."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if args.trio or args.coverage != 20 or args.path not in ("auto", "resident") or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["traffic_bytes_per_launch"]


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch  # device selection, synchronisation and the rendezvous only
    import __graft_entry__ as entry

    if not os.path.exists(os.path.join(ROOT, "whatshap_amd", "libwhatshap_amd.so")):
        entry.build()
    from whatshap_amd import _native
    from whatshap_amd.blocks import assign_blocks, block_weight
    from whatshap_amd.synthetic import synthetic_block

    if not torch.cuda.is_available() or _native.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank, to exercise the path)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # ---- the job: world * blocks_per_gpu independent blocks, assigned largest-first to the least loaded rank
    n_blocks = world * args.blocks_per_gpu
    seeds = [3 + b for b in range(n_blocks)]
    weights = [block_weight(args.variants, args.coverage, 4 if args.trio else 1) for _ in seeds]
    mine = assign_blocks(weights, world)[rank]
    tables = []
    for b in mine:
        problem = synthetic_block(args.variants, args.coverage, seed=seeds[b], trio=args.trio)
        tables.append(_native.NativeTable(problem, device=local_rank, path=None if args.path == "auto" else args.path,
                                          solve=False))

    def step():
        # host-side work queue: the blocks of this rank are submitted to their own streams (launch sequences
        # interleaved, so that they start together), then collected
        _native.enqueue_many(tables)
        for t in tables:
            t.wait()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    fwd_ms = bt_ms = total_ms = 0.0
    launches = 0
    for _ in range(args.steps):
        step()
        for t in tables:
            s = t.stats()
            fwd_ms += s["forward_ms"]
            bt_ms += s["backtrace_ms"]
            total_ms += s["total_ms"]
            launches += s["forward_launches"]
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    stats = [t.stats() for t in tables]
    cols_rank = sum(s["n_columns"] for s in stats)
    costs_rank = sum(s["n_costs"] for s in stats)
    bytes_rank = sum(s["algorithmic_bytes"] for s in stats)
    # every rank holds blocks of identical shape, so whole-job totals are world * per-rank totals
    cols_job = cols_rank * world
    costs_job = costs_rank * world
    checksum = sum(t.optimal_score() for t in tables)

    if rank == 0:
        # dominant kernel: the column step (one launch per column).  Algorithmic bytes per launch:
        # 4*T*2^b (read previous projection) + 12*T*2^f (projection + two backtrace tables) + 12*k.
        avg_launch_us = fwd_ms * 1e3 / max(launches, 1)
        bytes_per_launch = bytes_rank / max(launches / args.steps, 1)
        achieved = bytes_per_launch / (avg_launch_us * 1e-6) / 1e9
        out = {
            "metric": "variant-columns/sec at max-coverage %d (bipartition-costs/sec reported alongside)" % args.coverage,
            "value": cols_job * args.steps / elapsed,
            "unit": "variant-columns/s",
            "bipartition_costs_per_s": costs_job * args.steps / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": ("synthetic trio PedMEC ReadSet" if args.trio else "synthetic diploid single-individual ReadSet")
                            + f", {args.variants} het SNVs, max-coverage {args.coverage}, {args.blocks_per_gpu} block(s) per GPU"
                            + (" (BASELINE configs[2])" if (not args.trio and args.variants == 200000 and args.coverage == 20) else ""),
                "n_variants": args.variants,
                "max_coverage": args.coverage,
                "transmission_values": 4 if args.trio else 1,
                "blocks": n_blocks,
                "path": args.path,
                "optimal_cost_checksum_rank0": checksum,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("column_step_fused<%d,%d>" % ((4, 3) if args.trio else (1, 1))) if args.path in ("column", "column_keys")
                          else ("resident_segment_ped" if args.trio else "resident_segment"),
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": measured_traffic(args),
                "avg_launch_us": avg_launch_us,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "forward_ms_per_step": fwd_ms / args.steps,
                "backtrace_ms_per_step": bt_ms / args.steps,
            },
        }
        n_cpu = args.cpu_baseline_columns
        if n_cpu < 0:
            # bounded sample of ~15 s of CPU work.  Calibrated on two short prefixes (host CPUs differ by 2x): the first
            # 2 * coverage columns are the coverage ramp of the synthetic ReadSet and cost next to nothing
            ramp = 2 * args.coverage
            a, b = cpu_baseline(args, ramp + 8), cpu_baseline(args, ramp + 24)
            per_col = max((b["seconds"] - a["seconds"]) / 16.0, 1e-6)
            n_cpu = int(max(ramp + 24, min(args.variants, ramp + 8 + (15.0 - a["seconds"]) / per_col)))
        if n_cpu > 0:
            out["cpu_baseline"] = cpu_baseline(args, n_cpu)
            out["speedup_vs_cpu_baseline"] = out["value"] / world / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
