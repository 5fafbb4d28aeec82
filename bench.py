#!/usr/bin/env python3
"""Benchmark of the wMEC/PedMEC hot path (BASELINE.json metric) on 1..N MI355X.

A *step* is one complete pass of the hot path -- forward DP over every column + backtrace to the index path + host
result extraction -- over the rank's synthetic phasing block(s), with the flattened input already resident in HBM.

Workloads (BASELINE.json `configs`):
  N = 1 : configs[2] -- synthetic diploid single-individual ReadSet, 200 000 het SNVs, max-coverage 20 (2^20
          bipartitions per column), ONE table.
  N > 1 : configs[4] -- 24 independent blocks x 100 000 SNVs, max-coverage 20 (seeds 100..123), assigned longest-first
          to the least loaded rank (LPT), every rank keeps its blocks in flight through the host-side work queue
          (`whamd_dptable_enqueue_many`), no data-path collective; total work is fixed => "scaling": "strong".
  (`--blocks-per-gpu B` instead gives every rank B blocks of `--variants` columns: weak scaling, for A/B runs.)

`python bench.py --gpus N` starts its N ranks itself (one process per GPU through torch.distributed.run on
127.0.0.1) when it was not already launched by torchrun, and fails loudly when fewer than N devices are visible.

At N = 1 with no workload flags the line also carries `configs`: the other single-GPU workloads of BASELINE.json
(configs[1] 50 000 x coverage 15, configs[3] trio, three configs[4] blocks in flight) plus an irregular read layout
(Poisson starts, geometric lengths: the planner's typical case rather than its best) and a quartet, each measured by a child
`bench.py --workload NAME --sub` with its own value / ms_per_step / steps / roofline.frac / cpu_baseline sample.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      counter-based: the fraction of the chip's VALU issue slots the dominant kernel uses (the binding roof of
                this integer min-plus path; no MFMA), next to the LDS-pipe and MEASURED HBM fractions.  The counters
                come from rocprofv3 --pmc passes that bench.py itself runs after the timed region (separate passes,
                only with --kernel-trace) on a short slice of the same workload; `hbm_model_ratio` keeps SURVEY.md
                8(d)'s algorithmic-bytes figure, labelled as what it is (bytes the reference's tables would move).
  cpu_baseline  the compiled reference (oracle/_ref, 1 thread) on steady-state columns of the same ReadSet.
  end_to_end    a fresh table: create (flatten + plan + upload) + solve + getters, columns/s (host-inclusive).
"""

import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
CLOCK_GHZ = 2.4          # max shader clock
N_SIMD = 1024            # 256 CUs x 4 SIMD-32
N_CU = 256
CONFIG4_BLOCKS, CONFIG4_VARIANTS, CONFIG4_SEED0 = 24, 100000, 100
# named single-GPU workloads: flags they stand for
WORKLOADS = {
    "config1": dict(variants=50000, coverage=15),                                  # BASELINE configs[1]
    "config2": dict(variants=200000, coverage=20),                                 # BASELINE configs[2] (the headline)
    "config3": dict(trio=True, variants=100000, coverage=15),                      # BASELINE configs[3]
    "blocks3": dict(variants=100000, coverage=20, blocks=3, in_flight=3),          # three configs[4] blocks in flight on one GPU
    "blocks24": dict(variants=100000, coverage=20, blocks=24, in_flight=24, option=["shared_launches=1"]),       # BASELINE configs[4] on ONE GPU: all 24 blocks as one group of launches
    "config1_x24": dict(variants=50000, coverage=15, blocks=24, in_flight=24, option=["shared_launches=1"]),     # 24 tables at `whatshap phase`'s default coverage (24 chromosomes) on one GPU
    "config1_x48": dict(variants=50000, coverage=15, blocks=48, in_flight=48, option=["shared_launches=1"]),     # twice that: 384 workgroups per launch (does the launch time hold?)
    "config1_x96": dict(variants=50000, coverage=15, blocks=96, in_flight=96, option=["shared_launches=1"]),     # a cohort's worth of chromosome x family tables: three workgroups per CU per launch
    "config3_distrust": dict(trio=True, distrust=True, variants=100000, coverage=15),   # configs[3]'s ReadSet, genotypes not trusted (16 allele assignments per value)
    "config3_x8": dict(trio=True, variants=100000, coverage=15, blocks=8, in_flight=8),  # eight trio tables (families / chromosomes) on one GPU
    "irregular": dict(irregular=True, variants=100000, coverage=20),               # Poisson starts, geometric lengths (mean 16), coverage capped
    "irregular_x24": dict(irregular=True, variants=50000, coverage=15, blocks=24, in_flight=24, option=["shared_launches=1"]),   # what `whatshap phase` produces: default --internal-downsampling 15 (cli/phase.py:1066-1069), real read lengths, 24 chromosomes
    "irregular_cov20_x12": dict(irregular=True, variants=100000, coverage=20, blocks=12, in_flight=12, option=["shared_launches=1"]),   # twelve irregular coverage-20 tables sharing their launches
    "config_cov23": dict(variants=20000, coverage=23),                             # the CLI's cap (cli/phase.py:1181-1182): ONE table fills the chip -- 2 048 workgroups per launch
    "quartet": dict(quartet=True, variants=50000, coverage=13),                    # two trios sharing parents, T = 16
    "quartet_distrust": dict(quartet=True, distrust=True, variants=50000, coverage=13),   # the same, genotypes not trusted: the quartet's factorised lines (slots.h PSLOT_FACT4)
    "genotype": dict(genotype=True, variants=50000, coverage=15),                  # GenotypeDPTable, single individual
    "genotype_trio": dict(genotype=True, trio=True, variants=20000, coverage=15),  # GenotypeDPTable, trio
    "config2_shim": dict(shim=True, variants=200000, coverage=20),                 # configs[2] entered as `whatshap phase` enters: WhatsHap's own ReadSet / Pedigree in, its ReadSets out
    "config3_shim": dict(shim=True, trio=True, variants=100000, coverage=15),      # configs[3] the same way (three individuals' superreads)
    "heuristic": dict(heuristic=True, variants=8000, coverage=30),                 # PedMecHeuristic, coverage beyond the exact DP
    "heuristic_x32": dict(heuristic=True, variants=8000, coverage=30, blocks=32),  # 32 PedMecHeuristic tables in ONE launch (one persistent workgroup each)
}
EXTRA_CONFIGS = ["config2_shim", "config3_shim", "config1", "config1_x24", "config1_x96", "config3", "config3_distrust", "config3_x8", "blocks3", "blocks24", "irregular", "irregular_x24", "irregular_cov20_x12", "config_cov23", "quartet", "quartet_distrust", "genotype", "genotype_trio", "heuristic", "heuristic_x32"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--variants", type=int, default=None, help="columns per block (default: 200000 at N=1, 100000 at N>1)")
    ap.add_argument("--coverage", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=None, help="total independent blocks of the job (default: 1 at N=1, 24 at N>1)")
    ap.add_argument("--blocks-per-gpu", type=int, default=None, help="weak-scaling mode: every rank gets this many blocks")
    ap.add_argument("--in-flight", type=int, default=4, help="blocks a rank keeps in flight at once")
    ap.add_argument("--trio", action="store_true", help="configs[3]-shaped workload (trio PedMEC, coverage 15) instead")
    ap.add_argument("--quartet", action="store_true", help="two trios sharing their parents (T = 16), coverage 13")
    ap.add_argument("--distrust", action="store_true", help="distrust_genotypes=True: every allele assignment of an individual is allowed, priced by its genotype likelihood")
    ap.add_argument("--irregular", action="store_true", help="irregular read layout (whatshap_amd.synthetic.irregular_block, seed 7)")
    ap.add_argument("--genotype", action="store_true", help="the genotyping row: GenotypeDPTable (forward-backward, f64) instead of the phasing table")
    ap.add_argument("--heuristic", action="store_true", help="the PedMecHeuristic row: beam search at a coverage the exact DP cannot afford (row limit 256)")
    ap.add_argument("--shim", action="store_true", help="enter through whatshap_amd.shim with the reference's own ReadSet / Pedigree objects (the compiled whatshap.core of oracle/_ref/cy as the container types)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="a named workload (sets the flags above)")
    ap.add_argument("--configs", default="auto", choices=["auto", "on", "off"],
                    help="append the other single-GPU workloads as `configs` (auto: N=1 and no workload flags)")
    ap.add_argument("--sub", action="store_true", help=argparse.SUPPRESS)   # child of the `configs` array: short counters, short CPU sample, no torch
    ap.add_argument("--oversubscribe", action="store_true", help="allow more ranks than devices (rank r uses device r %% visible; tests)")
    ap.add_argument("--path", default="auto")
    ap.add_argument("--option", action="append", default=[], help="key=value passed to whamd_dptable_set_option")
    ap.add_argument("--cpu-baseline-columns", type=int, default=-1,
                    help="steady-state columns of the same ReadSet timed on the compiled reference (default: ~15 s worth; 0 = skip)")
    ap.add_argument("--cpu-baseline-procs", type=int, default=0,
                    help="also time P concurrent single-thread reference processes on independent blocks (BASELINE.md 3.5; -1 = nproc)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "on", "off"],
                    help="run the rocprofv3 counter passes after the timed region (auto: N=1 and rocprofv3 present)")
    ap.add_argument("--pmc-variants", type=int, default=8000)
    ap.add_argument("--pmc-keep", default=os.path.join(ROOT, "gpurun_out", "pmc_live"), help="where the filtered counter CSVs are kept")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"), help="where the full record (every entry, every counter) is written")
    ap.add_argument("--pmc-inner", action="store_true", help=argparse.SUPPRESS)     # the profiled child: solve and exit
    ap.add_argument("--cpu-sample-worker", type=int, default=0, help=argparse.SUPPRESS)  # child of --cpu-baseline-procs
    ap.add_argument("--heuristic-cpu-worker", type=int, default=0, help=argparse.SUPPRESS)  # child of the batched heuristic entry
    ap.add_argument("--create-rate-worker", default=None, help=argparse.SUPPRESS)   # "r/n": child of the create_rate entry, bound to the CPU slice of rank r of n
    ap.add_argument("--no-affinity", action="store_true", help="do not bind a rank's host threads to the CPU slice of its GPU (N > 1)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ workload
def resolve_workload(args, world):
    """-> (list of (seed, n_variants) for every block of the job, scaling, description)."""
    cov = args.coverage
    if args.trio and args.coverage == 20:
        cov = args.coverage = 15
    if args.quartet and args.coverage == 20:
        cov = args.coverage = 13
    if args.blocks_per_gpu is not None:
        v = args.variants or 200000
        blocks = [(3 + b, v) for b in range(world * args.blocks_per_gpu)]
        return blocks, "weak", f"{args.blocks_per_gpu} block(s) of {v} SNVs per GPU"
    if world == 1 and args.blocks is None:
        v = args.variants or (100000 if args.trio else (50000 if args.quartet else 200000))
        tag = ""
        if not args.trio and v == 200000 and cov == 20:
            tag = " (BASELINE configs[2])"
        if args.trio and v == 100000 and cov == 15:
            tag = " (BASELINE configs[3])"
        if args.irregular:
            return [(7, v)], "weak", f"1 block of {v} SNVs, irregular read layout (Poisson starts, geometric lengths of mean {0.8 * cov:.0f}, seed 7)"
        return [(4 if args.trio else (5 if args.quartet else 3), v)], "weak", f"1 block of {v} SNVs{tag}"
    n = args.blocks or CONFIG4_BLOCKS
    v = args.variants or CONFIG4_VARIANTS
    tag = " (BASELINE configs[4])" if (n == CONFIG4_BLOCKS and v == CONFIG4_VARIANTS and cov == 20 and not args.trio) else ""
    if args.irregular:
        tag = f", irregular read layout (Poisson starts, geometric lengths of mean {0.8 * cov:.0f})"
    return [(CONFIG4_SEED0 + b, v) for b in range(n)], "strong", f"{n} independent blocks x {v} SNVs, LPT over {world} GPU(s){tag}"


def build_block(args, seed, n_variants, n_columns_limit=None):
    """The seeded block of the selected workload (optionally only its first columns: the CPU-baseline samples)."""
    from whatshap_amd.synthetic import clip_to_columns, irregular_block, synthetic_block

    if args.irregular:
        p = irregular_block(n_variants, args.coverage, seed=seed)
        return p if n_columns_limit is None else clip_to_columns(p, n_columns_limit)
    return synthetic_block(n_variants, args.coverage, seed=seed, trio=args.trio, quartet=args.quartet, distrust_genotypes=args.distrust, n_columns_limit=n_columns_limit)


def workload_flags(args):
    out = []
    for flag in ("trio", "quartet", "distrust", "irregular", "genotype", "heuristic", "shim"):
        if getattr(args, flag):
            out.append("--" + flag)
    return out


def option_dict(args):
    return dict(kv.partition("=")[::2] for kv in args.option)


def apply_options(table, args):
    for kv in args.option:
        key, _, value = kv.partition("=")
        table.set_option(key, value)


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "model": model}


def solution_tuple(table):
    """Everything the drop-in class hands back (whatshap/core.pyx:364-416): cost, index path, transmission vector, partitioning, superreads."""
    a0, a1, q, tv, sid = table.super_reads()
    idx, tv2 = table.index_path()
    return (int(table.optimal_score()), idx.tolist(), tv.tolist(), tv2.tolist(), table.partitioning().tolist(), a0.tolist(), a1.tolist(), q.tolist(), sid.tolist())


def reference_seconds(args, seed, n_columns, want_solution=False):
    """Constructor + the three getters of the compiled reference (oracle/_ref; the C restatement if it is absent) on the
    first `n_columns` columns of the seeded ReadSet; single thread."""
    import oracle

    kind = "reference" if oracle.have_reference() else "port"
    table_cls = oracle.ReferenceTable if kind == "reference" else oracle.OracleTable
    problem = build_block(args, seed, args.variants_for_cpu, n_columns_limit=n_columns)
    t0 = time.perf_counter()
    table = table_cls(problem)
    score = table.optimal_score()
    table.super_reads()
    table.partitioning()
    seconds = time.perf_counter() - t0
    if want_solution:
        return seconds, table.n_columns, score, kind, solution_tuple(table), problem
    return seconds, table.n_columns, score, kind


def cpu_baseline(args, seed):
    """Steady-state columns/s of the single-thread reference: the first 2 * coverage columns of the synthetic ReadSet
    are its coverage ramp (next to free), so the rate is the DIFFERENCE between two prefixes that both contain it."""
    ramp = 2 * args.coverage
    n_cpu = args.cpu_baseline_columns
    ta, ca, _, kind = reference_seconds(args, seed, ramp + 8)
    done = None
    if n_cpu < 0:
        tb, cb, score_b, _, want_b, prefix_b = reference_seconds(args, seed, ramp + 24, want_solution=True)
        per_col = max((tb - ta) / max(cb - ca, 1), 1e-6)
        budget = 4.0 if args.sub else 15.0
        n_cpu = int(max(24, min(args.variants_for_cpu - ramp - 8, budget / per_col)))
        if budget / per_col <= (cb - ca) * 1.5:
            # wide columns (coverage 22-23: seconds per column on the reference): the second prefix already is a sample of the budget's size
            done = (tb, cb, score_b, kind, want_b, prefix_b)
    tc, cc, score, _, want, prefix = done if done is not None else reference_seconds(args, seed, ramp + 8 + n_cpu, want_solution=True)
    steady_cols, steady_s = cc - ca, max(tc - ta, 1e-9)
    info = cpu_info()
    # the parity bit of this line: the SAME prefix on the device (the product path, through the C ABI), full tuple compared
    from whatshap_amd import _native

    mine = _native.NativeTable(prefix, device=args.device_for_parity, path=None if args.path == "auto" else args.path, solve=False)
    apply_options(mine, args)
    mine.solve()
    identical = solution_tuple(mine) == want
    mine.close()
    return {
        "identical_to_reference": bool(identical),
        "identical_what": f"cost, index path, transmission vector, partitioning and superreads (alleles + qualities) of the first {cc} columns: device vs the {kind}",
        "value": steady_cols / steady_s,
        "unit": "variant-columns/s",
        "cores": 1,
        "kind": kind,
        "sample": f"columns {ca}..{cc} of the same seeded ReadSet (coverage {args.coverage}{', trio' if args.trio else ''}{', genotypes not trusted' if args.distrust else ''}{', quartet' if args.quartet else ''}{', irregular layout' if args.irregular else ''}; all at full "
                  f"coverage: the {ramp}-column ramp is timed separately and subtracted), constructor + 3 getters, {steady_s:.1f} s of "
                  f"{tc:.1f} s wall, optimal cost of the prefix {score}",
        "seconds": tc + ta,
        "host": info,
    }


def cpu_baseline_procs(args, procs):
    """P single-thread reference processes at once, each on its own block prefix (independent blocks
    are the only parallelism the reference has).  Aggregate steady-state columns/s."""
    ramp = 2 * args.coverage
    cols = 16
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-sample-worker", str(cols), "--coverage", str(args.coverage),
           "--variants", str(args.variants_for_cpu)] + workload_flags(args)
    t0 = time.perf_counter()
    children = [subprocess.Popen(cmd + ["--blocks", str(CONFIG4_SEED0 + i)], stdout=subprocess.PIPE, text=True) for i in range(procs)]
    rates = []
    for ch in children:
        out, _ = ch.communicate(timeout=600)
        if ch.returncode == 0 and out.strip():
            rates.append(float(out.strip().splitlines()[-1]))
    return {"value": sum(rates), "unit": "variant-columns/s", "cores": len(rates), "kind": "reference",
            "sample": f"{len(rates)} concurrent single-thread processes, each {cols} steady-state columns (after the {ramp}-column ramp) "
                      f"of its own seeded block, {time.perf_counter() - t0:.1f} s wall", "host": cpu_info()}


def cpu_sample_worker(args):
    args.variants_for_cpu = args.variants or 100000
    seed = args.blocks or CONFIG4_SEED0
    ramp = 2 * args.coverage
    ta, ca, _, _ = reference_seconds(args, seed, ramp + 4)
    tb, cb, _, _ = reference_seconds(args, seed, ramp + 4 + args.cpu_sample_worker)
    # steady-state rate; with hundreds of processes contending, the short ramp-only run can be delayed by more than the
    # difference: never credit more than 4x the rate of the long run taken as a whole
    print((cb - ca) / max(tb - ta, 0.25 * tb * (cb - ca) / max(cb, 1)))


# ------------------------------------------------------------------------------------------------ counters
PMC_PASSES = [
    ("insts", "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"),
    ("fetch", "FETCH_SIZE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"),
    ("write", "WRITE_SIZE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"),
]


def run_pmc_passes(args, kernel_substring, keep_dir):
    """Runs the counter passes on `--pmc-variants` columns of the same workload (one rocprofv3 invocation per counter
    group, --pmc only ever combined with --kernel-trace) and returns per-dispatch averages over the FULL-WIDTH dispatches
    of the dominant kernel (the largest grid it was launched with)."""
    import csv
    import collections

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    inner = [sys.executable, os.path.abspath(__file__), "--pmc-inner", "--variants", str(args.pmc_variants), "--coverage", str(args.coverage),
             "--path", args.path, "--steps", "1", "--warmup", "1", "--configs", "off"] + workload_flags(args)
    if args.blocks:
        inner += ["--blocks", str(args.blocks), "--in-flight", str(args.in_flight)]
    for kv in args.option:
        inner += ["--option", kv]
    env = dict(os.environ, TMPDIR="/tmp")
    averages, counts, notes = {}, {}, []
    os.makedirs(keep_dir, exist_ok=True)
    passes = PMC_PASSES[:1] if args.sub else PMC_PASSES
    if args.genotype:   # how many of the VALU instructions are f64 (they hold a SIMD for 4 cycles, not 2)
        passes = list(passes) + [("f64", "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64")]
    for name, counters in passes:
        out_dir = tempfile.mkdtemp(prefix=f"whamd_pmc_{name}_", dir="/tmp")
        cmd = [rocprof, "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", out_dir, "-o", "p", "--"] + inner
        try:
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
        except subprocess.TimeoutExpired:
            notes.append(f"{name}: timeout")
            continue
        files = []
        for base, _, names in os.walk(out_dir):
            files += [os.path.join(base, n) for n in names if n.endswith("counter_collection.csv")]
        if res.returncode != 0 or not files:
            notes.append(f"{name}: rc={res.returncode} {res.stderr[-200:]}")
            continue
        rows = []
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if kernel_substring in row["Kernel_Name"]:
                    rows.append(row)
        if not rows:
            notes.append(f"{name}: kernel {kernel_substring} not in the trace")
            continue
        full = max(int(r["Grid_Size"]) for r in rows)
        sums, cnt = collections.Counter(), collections.Counter()
        kept = []
        for r in rows:
            if int(r["Grid_Size"]) != full:
                continue
            sums[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
            kept.append(r)
        for c in sums:
            averages[c] = sums[c] / cnt[c]
            counts[c] = cnt[c]
        averages["_grid_size"] = full
        averages["_workgroup_size"] = int(kept[0]["Workgroup_Size"])
        names = collections.Counter(r["Kernel_Name"] for r in kept)
        averages["_kernel_name"] = names.most_common(1)[0][0]
        with open(os.path.join(keep_dir, f"{name}_counter_collection.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(kept[0].keys()))
            w.writeheader()
            w.writerows(kept)
        shutil.rmtree(out_dir, ignore_errors=True)
    if not averages:
        return None, "; ".join(notes)
    averages["_dispatches"] = max(counts.values()) if counts else 0
    return averages, "; ".join(notes)


def roofline_from_counters(pmc, avg_launch_us, kernel, bytes_per_launch_model, args):
    """Counter-based fractions of the chip's peaks for the dominant kernel.  Durations are the
    un-profiled HIP-event launch time of the timed region; counters are per-dispatch averages of the profiled slice."""
    cycles = avg_launch_us * 1e-6 * CLOCK_GHZ * 1e9
    out = {
        "bound": "valu_issue",
        "kernel": kernel,
        "unit": "wave-instructions/cycle (chip)",
        "peak": N_SIMD / 2.0,   # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles
        "avg_launch_us": avg_launch_us,
        "clock_ghz_assumed": CLOCK_GHZ,
    }
    if pmc is None:
        out.update({"achieved": None, "frac": None, "traffic": None})
        return out
    if pmc.get("_kernel_name"):
        # the instantiation the counters were taken from (`kernel` was the filter: "slot_run" also matches slot_runx<2, 24, false, false>)
        out["kernel_filter"] = kernel
        name = str(pmc["_kernel_name"])
        out["kernel"] = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("whamd::", "").split("(")[0].strip() or kernel   # ("(anonymous namespace)" goes first: its bracket is not the argument list's)
    valu = pmc.get("SQ_INSTS_VALU")
    if valu is not None:
        out["achieved"] = valu / cycles
        out["frac"] = out["achieved"] / out["peak"]
        out["valu_issue_frac"] = out["frac"]
    act = pmc.get("SQ_ACTIVE_INST_VALU")
    if act is not None:
        # measured, not modelled: SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD's vector ALU is held by an instruction (the guide: SQ_WAVE_CYCLES /
        # SQ_ACTIVE_INST_* count quad-cycles).  Simple adds hold it for one, v_sad_u32 / v_min3_u32 / v_max_u32 for nearly two (scripts/micro/op_rates.hip)
        out["valu_active_frac"] = 4.0 * act / (N_SIMD * cycles)
        out["valu_active_note"] = "4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x launch cycles): the share of the launch in which the vector ALUs are held"
    f64 = [pmc.get(k) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")]
    if valu is not None and any(x is not None for x in f64):
        n64 = sum(x for x in f64 if x is not None)
        out["f64_instructions_per_launch"] = n64
        # SIMD-cycles the VALU instructions occupy (2 per wave64 instruction, 4 when it is f64) over the SIMD-cycles of the launch
        out["valu_busy_frac_f64_weighted"] = (2.0 * valu + 2.0 * n64) / (N_SIMD * cycles)
    fetch_kb, write_kb = pmc.get("FETCH_SIZE"), pmc.get("WRITE_SIZE")
    if fetch_kb is not None and write_kb is not None:
        # MI355X_MICROARCH.md "HBM": gfx950's FETCH_SIZE reports half of a wide coalesced read stream -> doubled; WRITE_SIZE as is
        traffic = 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0
        out["traffic"] = traffic
        out["hbm_frac"] = traffic / (avg_launch_us * 1e-6) / (HBM_PEAK_GBPS * 1e9)
        out["traffic_note"] = "bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB counters of separate rocprofv3 --pmc passes)"
    else:
        out["traffic"] = None
    lds = pmc.get("SQ_LDS_IDX_ACTIVE")
    if lds is not None:
        out["lds_pipe_frac"] = lds / (N_CU * cycles)
    out["hbm_model_ratio"] = bytes_per_launch_model / (avg_launch_us * 1e-6) / (HBM_PEAK_GBPS * 1e9)
    out["hbm_model_note"] = ("SURVEY.md 8(d) algorithmic bytes (4*T*2^b + 12*T*2^f + 12*k per column: the reference's three tables) per "
                             "launch / launch time / 8 TB/s -- NOT a roofline fraction: those tables are never materialised here")
    out["counters_per_launch"] = {k: v for k, v in pmc.items() if not k.startswith("_")}
    out["counters_kernel_name"] = pmc.get("_kernel_name")
    out["counters_measured_on"] = (f"{pmc.get('_dispatches')} full-width dispatches ({pmc.get('_grid_size')} work-items, "
                                   f"{pmc.get('_workgroup_size')} per workgroup) of `bench.py --variants {args.pmc_variants}` "
                                   f"(same seed, coverage {args.coverage}), rocprofv3 --kernel-trace --pmc, one pass per counter group")
    return out


# ------------------------------------------------------------------------------------------------ ranks
def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves."""
    from whatshap_amd import _native

    visible = _native.device_count()
    if visible < args.gpus and not args.oversubscribe:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {visible} HIP device(s) visible; refusing to run fewer ranks than asked for")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def with_uniform_priors(p):
    """Genotype priors 1/3, 1/3, 1/3 for every individual and column: what `whatshap genotype` passes when it has none."""
    import numpy as np
    from whatshap_amd import _native

    n_ind, n_var = p.n_individuals, p.n_variants
    gl = np.full((n_ind, n_var, 3), 1.0 / 3.0)
    return _native.ProblemArrays(p.read_ptr, p.var_position, p.var_allele, p.var_quality, p.read_sample_id, p.individual_id, p.triple_ids,
                                 p.genotype.reshape(n_ind, n_var), gl, p.recombcost, p.positions, False, n_variants=n_var)


def genotype_main(args):
    """`--genotype`: the genotyping row.  A step = one whamd_genotype_likelihoods call from host arrays
    (constructor + every likelihood of the reference class, whatshap/core.pyx:581-602); `value` = columns / the HIP-event
    time of a step (tables + both chains + combine; inputs uploaded before the events start, like the phasing line);
    `end_to_end` = columns / the wall time of the call.  roofline: the run kernel's VALU issue fraction (f64 instructions
    occupy a SIMD for 4 cycles instead of 2: `f64_weighted_frac`) and the combine kernel's measured HBM fraction."""
    import numpy as np
    from whatshap_amd import _native

    if _native.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.trio and args.coverage == 20:
        args.coverage = 15
    v = args.variants or (20000 if args.trio else 50000)
    seed = 4 if args.trio else 2
    problem = with_uniform_priors(build_block(args, seed, v))
    n = int(problem.positions.size)
    for _ in range(args.warmup):
        _native.genotype_likelihoods(problem, n)
    if args.pmc_inner:
        _native.genotype_likelihoods(problem, n)
        return
    wall, dev, stats = [], [], None
    for _ in range(args.steps):
        t0 = time.perf_counter()
        gl, stats = _native.genotype_likelihoods(problem, n)
        wall.append(time.perf_counter() - t0)
        dev.append(stats["total_ms"] / 1e3)
    dev_s, wall_s = float(np.median(dev)), float(np.median(wall))
    T = 4 if args.trio else 1
    out = {
        "metric": "variant-columns/sec of GenotypeDPTable (forward-backward + every genotype likelihood) at max-coverage %d" % args.coverage,
        "value": n / dev_s, "unit": "variant-columns/s", "cells_per_s": stats["n_cells"] * T / dev_s,
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dev_s * 1e3, "ms_per_step_min": min(dev) * 1e3, "ms_per_step_median": dev_s * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic {'trio' if args.trio else 'single individual'}, {n} SNVs, max-coverage {args.coverage}, uniform genotype priors, GenotypeDPTable",
                   "transmission_values": stats["transmissions"], "slot_runs_per_chain": stats["slot_runs"], "launches": stats["launches"],
                   "optimal_cost_checksum": int(round(float(gl[:, :, 1].sum()) * 1e6)),   # (sum of the heterozygous likelihoods x 1e6: a fingerprint of the output)
                   "value_is": "median over the steps of columns / HIP-event time of one call (tables + both chains + combine)"},
        "rank0": {"forward_ms_per_step": stats["backward_ms"], "backtrace_ms_per_step": stats["forward_ms"], "forward_launches_per_step": float(stats["launches"]),
                  "note": "forward_ms = the two chains side by side, backtrace_ms = tables + combine (the fields of the phasing line reused)"},
        "end_to_end": {"value": n / wall_s, "unit": "variant-columns/s", "ms_per_step": wall_s * 1e3,
                       "what": "whamd_genotype_likelihoods from host arrays: flatten + model + plan + upload + device + download (wall, median)"},
        "bipartition_costs_per_s": stats["n_cells"] * T / dev_s,
    }
    kernel = "geno_slot_run" if stats["slot_runs"] else "geno_forward"
    avg_launch_us = stats["backward_ms"] * 1e3 / max(stats["slot_runs"], 1) if stats["slot_runs"] else stats["total_ms"] * 1e3 / max(stats["launches"], 1)
    pmc, pmc_note = None, "skipped"
    if args.pmc in ("on", "auto"):
        _native.release_caches()
        try:
            pmc, pmc_note = run_pmc_passes(args, kernel, args.pmc_keep)
        except Exception as exc:  # noqa: BLE001
            pmc_note = repr(exc)
    roof = roofline_from_counters(pmc, avg_launch_us, kernel, 0.0, args)
    roof.pop("hbm_model_ratio", None)
    roof.pop("hbm_model_note", None)
    roof["pmc_note"] = pmc_note
    roof["launch_time_note"] = "chain time / runs per chain (the two chains run side by side on two streams: a launch's own duration is at most this)"
    out["roofline"] = roof
    if args.cpu_baseline_columns != 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from genotype_cases import reference_likelihoods
        from oracle import build_cython_ref

        if build_cython_ref.available():
            ref = build_cython_ref.import_reference()
            ramp = 2 * args.coverage
            cols = args.cpu_baseline_columns if args.cpu_baseline_columns > 0 else (12 if args.trio else (150 if args.sub else 400))
            times = {}
            for c in (ramp, ramp + cols):   # steady-state columns = the difference of two prefixes of the same ReadSet
                prefix = with_uniform_priors(build_block(args, seed, v, n_columns_limit=c))
                t0 = time.perf_counter()
                want = reference_likelihoods(prefix, ref)
                times[c] = time.perf_counter() - t0
            got, _ = _native.genotype_likelihoods(prefix, int(prefix.positions.size))
            steady = max(times[ramp + cols] - times[ramp], 1e-9)
            out["cpu_baseline"] = {"value": cols / steady, "unit": "variant-columns/s", "cores": 1, "kind": "reference",
                                   "sample": f"{cols} steady-state columns (prefix of {ramp + cols} minus prefix of {ramp} columns of the same ReadSet), whatshap.core.GenotypeDPTable "
                                             f"constructor + every get_genotype_likelihoods (long double), {times[ramp + cols]:.1f} s",
                                   "host": cpu_info()}
            out["parity_prefix_max_abs_diff"] = float(np.abs(got - want).max())
            # f64 on the device against the reference's long double: the tolerance of tests/test_gpu_genotype.py
            out["identical_to_reference"] = bool(np.allclose(got, want, rtol=1e-9, atol=1e-13))
            out["identical_what"] = "every genotype likelihood of the prefix within rtol 1e-9 / atol 1e-13 of whatshap.core.GenotypeDPTable (long double)"
            out["speedup_vs_cpu_baseline_device_only"] = out["value"] / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_baseline"] = out["end_to_end"]["value"] / out["cpu_baseline"]["value"]
    emit(out, args)
    if out.get("identical_to_reference") is False:
        sys.exit(3)


def heuristic_main(args):
    """`--heuristic`: PedMecHeuristic.  A step = one solve (constructor + solve() of the reference class) of every table of
    the workload -- `--blocks B` tables go out as ONE launch with one persistent workgroup each (whamd_pedmec_heuristic_enqueue_many);
    `value` = columns of all tables / HIP-event time of the launch; the CPU baseline is the compiled reference's solve() on a prefix of the
    same ReadSet (and, for several tables, the same on `nproc` concurrent processes: independent tables are the reference's only
    parallelism), and every output of that prefix is compared (`identical_to_reference`)."""
    from whatshap_amd import _native

    if _native.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.coverage == 20:
        args.coverage = 30
    v = args.variants or 8000
    n_tables = args.blocks or 1
    row_limit = 256
    problems = [build_block(args, 3 + i, v) for i in range(n_tables)]

    def solve_all():
        if n_tables == 1:
            return [_native.pedmec_heuristic(problems[0], row_limit=row_limit)]
        return _native.pedmec_heuristic_many(problems, row_limit=row_limit)

    for _ in range(args.warmup):
        solve_all()
    if args.pmc_inner:
        solve_all()
        return
    dev, wall, got = [], [], None
    for _ in range(args.steps):
        t0 = time.perf_counter()
        got = solve_all()
        wall.append(time.perf_counter() - t0)
        dev.append(got[0]["stats"]["device_ms"] / 1e3)
    dev_s, wall_s = sorted(dev)[len(dev) // 2], sorted(wall)[len(wall) // 2]
    cols = v * n_tables
    out = {
        "metric": "variant-columns/sec of PedMecHeuristic.solve at max-coverage %d, row limit %d" % (args.coverage, row_limit),
        "value": cols / dev_s, "unit": "variant-columns/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dev_s * 1e3,
        "ms_per_step_min": min(dev) * 1e3, "ms_per_step_median": dev_s * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "bipartition_costs_per_s": sum(g["stats"]["total_solutions"] for g in got) / dev_s,
        "bipartition_costs_note": "partial solutions of the beam scored per second (sum over the columns of the beam's width), not table cells",
        "config": {"workload": f"synthetic single individual, {n_tables} table(s) x {v} SNVs, max-coverage {args.coverage}, PedMecHeuristic row limit {row_limit}"
                               + (", ONE launch with one persistent workgroup per table" if n_tables > 1 else ""),
                   "optimal_cost_checksum": sum(int(g["bipartition"].sum()) * 1000003 + int(g["transmission"].sum()) for g in got),
                   "widest_column": max(g["stats"]["max_solutions"] for g in got), "blocks_in_flight_per_gpu": n_tables, "tables_per_launch": n_tables},
        "rank0": {"forward_ms_per_step": dev_s * 1e3, "backtrace_ms_per_step": 0.0, "forward_launches_per_step": 1.0},
        "end_to_end": {"value": cols / wall_s, "unit": "variant-columns/s", "what": "plans + upload + kernel + phasing of every table from host arrays, wall"},
    }
    pmc, pmc_note = None, "skipped"
    if args.pmc in ("on", "auto"):
        try:
            pmc, pmc_note = run_pmc_passes(args, "heuristic_kernel", args.pmc_keep)
        except Exception as exc:  # noqa: BLE001
            pmc_note = repr(exc)
    roof = roofline_from_counters(pmc, dev_s * 1e6, "heuristic_kernel", 0.0, args)
    roof.pop("hbm_model_ratio", None)
    roof.pop("hbm_model_note", None)
    roof["pmc_note"] = pmc_note
    roof["shape_note"] = (f"{n_tables} persistent workgroup(s), one per table (a chain over columns and reads; the beam is the only parallelism inside a table): "
                          f"the chip-wide fraction is bounded by {n_tables} / 256 CUs")
    out["roofline"] = roof
    if args.cpu_baseline_columns != 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle
        from heuristic_cases import result_tuple

        if oracle.have_reference():
            cols_cpu = args.cpu_baseline_columns if args.cpu_baseline_columns > 0 else (2000 if args.sub else 6000)
            prefix = build_block(args, 3, v, n_columns_limit=min(cols_cpu, v))
            ref = oracle.ReferenceHeuristic(prefix, row_limit=row_limit)
            mine = _native.pedmec_heuristic(prefix, row_limit=row_limit)
            out["cpu_baseline"] = {"value": prefix.n_variants / ref.solve_seconds(), "unit": "variant-columns/s", "cores": 1, "kind": "reference",
                                   "sample": f"PedMecHeuristic::solve() of the compiled reference on the first {prefix.n_variants} columns of the same ReadSet, {ref.solve_seconds():.2f} s",
                                   "host": cpu_info()}
            out["identical_to_reference"] = result_tuple(mine) == oracle.heuristic_tuple(ref)
            out["speedup_vs_cpu_baseline_device_only"] = out["value"] / out["cpu_baseline"]["value"]
            if n_tables > 1:
                # the honest comparison for independent tables: the reference on every host core at once, one table prefix per process
                procs = os.cpu_count() or 1
                cmd = [sys.executable, os.path.abspath(__file__), "--heuristic-cpu-worker", str(min(1500, v)), "--coverage", str(args.coverage), "--variants", str(v)]
                t0 = time.perf_counter()
                children = [subprocess.Popen(cmd + ["--blocks", str(3 + (i % n_tables))], stdout=subprocess.PIPE, text=True) for i in range(procs)]
                rates = []
                for ch in children:
                    o, _ = ch.communicate(timeout=900)
                    if ch.returncode == 0 and o.strip():
                        rates.append(float(o.strip().splitlines()[-1]))
                out["cpu_baseline_all_cores"] = {"value": sum(rates), "unit": "variant-columns/s", "cores": len(rates), "kind": "reference",
                                                 "sample": f"{len(rates)} concurrent single-thread processes, each PedMecHeuristic::solve() on the first {min(1500, v)} columns of one of the "
                                                           f"{n_tables} ReadSets, {time.perf_counter() - t0:.1f} s wall", "host": cpu_info()}
                out["speedup_vs_cpu_all_cores_device_only"] = out["value"] / max(sum(rates), 1e-9)
    emit(out, args)
    if out.get("identical_to_reference") is False:
        sys.exit(3)


def shim_main(args):
    """`--shim`: the drop-in as `whatshap phase` sees it.  WhatsHap's OWN `ReadSet` / `Pedigree` objects go in (the compiled
    `whatshap.core` of oracle/_ref/cy provides the container types -- nothing of it computes here), `shim.install` rebinds a phase-like module,
    and a step is what `cli/phase.py:604-612` does with the table: constructor (compiled ingestion through thisptr + whamd_dptable_create + solve),
    `get_super_reads()` (reference ReadSets, emitted in C++ and adopted), `get_optimal_cost()`, `get_optimal_partitioning()`.  `value` = columns / median
    wall time of a step; `native_end_to_end` = the same table from host arrays through the C ABI (create + solve + 3 getters) in the same process."""
    import types

    from whatshap_amd import _native, ingest, shim
    from oracle import build_cython_ref

    if _native.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if not build_cython_ref.available():
        raise SystemExit("bench.py --shim needs the reference's compiled whatshap.core (oracle/_ref/cy) as the container types")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from refobjects import problem_to_reference, table_outputs

    ref = build_cython_ref.import_reference()
    if args.trio and args.coverage == 20:
        args.coverage = 15
    v = args.variants or (100000 if args.trio else 200000)
    seed = 4 if args.trio else 3
    problem = build_block(args, seed, v)
    rs, ped = problem_to_reference(problem, ref)
    recomb, positions = problem.recombcost.tolist(), problem.positions.tolist()     # Python lists, as cli/phase.py passes them
    phase = types.SimpleNamespace(Pedigree=ref.Pedigree, PedigreeDPTable=ref.PedigreeDPTable)
    shim.install(phase, ref)
    shim.reset_stats()

    def step():
        t0 = time.perf_counter()
        table = phase.PedigreeDPTable(rs, recomb, ped, args.distrust, positions)
        t1 = time.perf_counter()
        sets, tv = table.get_super_reads()
        t2 = time.perf_counter()
        cost = table.get_optimal_cost()
        part = table.get_optimal_partitioning()
        t3 = time.perf_counter()
        st = table._table.get_stats()
        return (t3 - t0, t1 - t0, t2 - t1, t3 - t2), cost, st

    def native_step():
        t0 = time.perf_counter()
        t = _native.NativeTable(problem, solve=False)
        t1 = time.perf_counter()
        t.solve()
        t.optimal_score(), t.super_reads(), t.partitioning()
        t2 = time.perf_counter()
        t.close()
        return t2 - t0, t1 - t0

    for _ in range(max(args.warmup, 1)):
        step()
        native_step()
    if args.pmc_inner:
        step()
        return
    walls, natives, st, cost = [], [], None, 0
    for _ in range(args.steps):
        w, cost, st = step()
        walls.append(w)
        natives.append(native_step())
    med = sorted(walls)[len(walls) // 2]
    nat = sorted(natives)[len(natives) // 2]
    T = 4 if args.trio else 1
    kernel = dominant_kernel(args)
    out = {
        "metric": "variant-columns/sec at max-coverage %d through whatshap_amd.shim: reference ReadSet / Pedigree objects in, reference ReadSets out" % args.coverage,
        "value": v / med[0], "unit": "variant-columns/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": med[0] * 1e3,
        "ms_per_step_min": min(w[0] for w in walls) * 1e3, "ms_per_step_median": med[0] * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "bipartition_costs_per_s": st["n_costs"] / med[0],
        "config": {"workload": f"synthetic {'trio PedMEC' if args.trio else 'diploid single-individual'}, {v} SNVs, max-coverage {args.coverage}, entered through whatshap_amd.shim.install with "
                               f"the reference's own ReadSet ({len(rs)} reads) / Pedigree objects; a step = constructor + get_super_reads + get_optimal_cost + get_optimal_partitioning",
                   "optimal_cost_checksum": int(cost), "transmission_values": T, "max_coverage": args.coverage,
                   "compiled_ingestion": ingest.load() is not None, "shim_stats": shim.stats()},
        "rank0": {"forward_ms_per_step": st["forward_ms"], "backtrace_ms_per_step": st["backtrace_ms"], "forward_launches_per_step": float(st["forward_launches"])},
        "step_pieces_ms": {"constructor": med[1] * 1e3, "get_super_reads": med[2] * 1e3, "cost_and_partitioning": med[3] * 1e3, "device_total": st["total_ms"],
                           "host_prepare": st["host_prepare_ms"]},
        "native_end_to_end": {"value": v / nat[0], "ms": nat[0] * 1e3, "create_ms": nat[1] * 1e3,
                              "what": "the same table from host arrays through the C ABI in the same process: whamd_dptable_create + solve + 3 getters (arrays out)"},
        "end_to_end": {"value": v / med[0], "wall_ms": med[0] * 1e3, "fraction_of_device_only": (v / med[0]) / (v / (st["total_ms"] * 1e-3)),
                       "what": "the step itself: reference objects in, reference objects out"},
        "shim_over_native": med[0] / nat[0],
    }
    roof = roofline_from_counters(None, st["forward_ms"] * 1e3 / max(st["forward_launches"], 1), kernel, 0.0, args)
    roof.pop("hbm_model_ratio", None)
    roof["pmc_note"] = "the device work is the headline's (config2) / config3's: counters there"
    out["roofline"] = roof
    if args.cpu_baseline_columns != 0:
        # the reference CLASS on a prefix of the same objects' data: the CPU rate of the step, and the parity bit -- objects compared with objects
        ramp = 2 * args.coverage
        times = {}
        for cols in (ramp + 8, ramp + 8 + (args.cpu_baseline_columns if args.cpu_baseline_columns > 0 else (60 if args.trio else 40))):
            prefix = build_block(args, seed, v, n_columns_limit=cols)
            prs, pped = problem_to_reference(prefix, ref)
            pr, pp = prefix.recombcost.tolist(), prefix.positions.tolist()
            t0 = time.perf_counter()
            want = table_outputs(ref.PedigreeDPTable(prs, pr, pped, args.distrust, pp))
            times[cols] = time.perf_counter() - t0
        got = table_outputs(phase.PedigreeDPTable(prs, pr, pped, args.distrust, pp))
        lo, hi = sorted(times)
        out["cpu_baseline"] = {"value": (hi - lo) / max(times[hi] - times[lo], 1e-9), "unit": "variant-columns/s", "cores": 1, "kind": "reference",
                               "sample": f"whatshap.core.PedigreeDPTable (the compiled reference class) constructor + 3 getters on columns {lo}..{hi} of the same data as reference objects "
                                         f"(steady state: prefix of {hi} minus prefix of {lo}), {times[hi]:.1f} s", "host": cpu_info()}
        out["identical_to_reference"] = got == want
        out["identical_what"] = f"cost, partitioning, transmission vector and the superread OBJECTS (names, sample ids, source ids, mapqs, every position / allele / quality) of the first {hi} columns: shim vs the reference class"
        out["speedup_vs_cpu_baseline_device_only"] = out["value"] / out["cpu_baseline"]["value"]
    emit(out, args)
    if out.get("identical_to_reference") is False:
        sys.exit(3)


def create_rate_worker(args):
    """Child of the `create_rate` entry: binds itself to the CPU slice rank r of n would get (BEFORE the library sizes its workers), then times
    whamd_dptable_create of every table of the workload on a pool of host workers -- no solve; prints one JSON object."""
    from concurrent.futures import ThreadPoolExecutor

    from whatshap_amd import _native
    from whatshap_amd.blocks import bind_rank_to_device_cpus

    r, _, n = args.create_rate_worker.partition("/")
    visible = max(_native.device_count(), 1)
    binding = bind_rank_to_device_cpus(int(r), int(n), devices=[i % visible for i in range(int(n))], spread_nodes=visible < int(n))
    n_cpus = min(len(os.sched_getaffinity(0)), int(binding.get("cpu_quota") or 1 << 30) or 1)   # (the slice of rank r of n; THIS box's quota, if it has one, is all the child can use -- what an 8-GPU node grants its ranks is not known here)
    n_tables = args.blocks or 1
    problems = [build_block(args, CONFIG4_SEED0 + i, args.variants or CONFIG4_VARIANTS) for i in range(n_tables)]
    native_path = None if args.path == "auto" else args.path
    best = None
    shapes = list(dict.fromkeys([(max(1, min(n_tables, n_cpus // 2)), 2), (max(1, min(n_tables, n_cpus // 4)), 4), (max(1, min(n_tables, n_cpus)), 1)]))
    for rep in range(2):    # (the first round also sizes the pools)
        for workers, per_create in shapes:
            opts = dict(option_dict(args), host_threads=str(per_create))
            with ThreadPoolExecutor(max_workers=workers) as pool:
                t0 = time.perf_counter()
                made = list(pool.map(lambda pr: _native.NativeTable(pr, device=0, path=native_path, solve=False, options=opts), problems))
                wall = time.perf_counter() - t0
            for t in made:
                t.close()
            if rep and (best is None or wall < best[0]):
                best = (wall, workers, per_create)
    print(json.dumps({"tables_per_s": n_tables / best[0], "create_wall_ms": best[0] * 1e3, "create_threads": best[1], "host_threads_per_create": best[2],
                      "thread_ms_per_table": best[0] * 1e3 * min(best[1] * best[2], n_cpus) / n_tables, "cpus": n_cpus, "cpu_source": binding.get("source"), "numa_node": binding.get("node")}))


def heuristic_cpu_worker(args):
    """Child of heuristic_main: the compiled reference's solve() on a prefix of one seeded ReadSet; prints columns/s."""
    import oracle

    args.coverage = 30 if args.coverage == 20 else args.coverage
    prefix = build_block(args, args.blocks or 3, args.variants or 8000, n_columns_limit=args.heuristic_cpu_worker)
    ref = oracle.ReferenceHeuristic(prefix, row_limit=256)
    print(prefix.n_variants / ref.solve_seconds())


# ------------------------------------------------------------------------------------------------ the line the driver parses
LINE_LIMIT = 6000     # bytes; the driver's record keeps a bounded tail of stdout and its parser gave up on round 4's 22.9 KB line
HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")


def _num(x, digits=4):
    """Numbers of the compact line: whole numbers as ints, the rest to `digits` significant digits."""
    if isinstance(x, bool) or x is None or isinstance(x, str):
        return x
    if isinstance(x, int):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    if abs(x) >= 10 ** digits:
        return int(round(x))
    return float(f"{x:.{digits}g}")


def _short(text, limit):
    """At most `limit` characters, cut at a word boundary (never mid-word) with a trailing ellipsis mark."""
    text = " ".join(str(text).split())
    if len(text) <= limit:
        return text
    cut = text[:limit - 2]
    if " " in cut[limit // 2:]:
        cut = cut[:cut.rindex(" ")]
    return cut.rstrip(" ,;:(") + " ~"


def compact_line(out, detail_file=None):
    """The LAST stdout line: the contract's keys, the headline's roofline and cpu_baseline, and ONE short record per entry of `configs`
    (value, ms_per_step, the dominant kernel's VALU issue fraction and launch time, the CPU rate, the parity bit, end to end).  Everything else --
    counters, notes, samples, host shapes tried -- is printed before it as `# bench detail` lines and written to `detail_file`."""
    line = {k: (_num(out[k], 7) if k in ("value", "ms_per_step") else out[k]) for k in HEAD_KEYS if k in out}
    line["metric"] = _short(line.get("metric", ""), 110)
    cfg = out.get("config", {})
    line["config"] = {"workload": _short(cfg.get("workload", ""), 150)}
    for k in ("blocks", "blocks_per_rank", "block_seeds_per_rank", "blocks_in_flight_per_gpu", "tables_per_launch", "max_coverage", "transmission_values", "path", "optimal_cost_checksum",
              "optimal_cost_checksum_per_rank", "device_per_rank", "rendezvous", "cpu_binding"):   # (what a SCALE record is audited with: which rank ran which blocks on which device)
        if k in cfg and len(json.dumps(cfg[k])) <= 160:
            line["config"][k] = cfg[k]
    roof = out.get("roofline")
    if roof:
        line["roofline"] = {k: (_num(roof[k]) if not isinstance(roof[k], str) else roof[k])
                            for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "valu_active_frac", "work_bound_frac", "traffic", "hbm_frac", "lds_pipe_frac", "avg_launch_us")
                            if k in roof}
        if roof.get("pmc_note"):
            line["roofline"]["pmc_note"] = _short(roof["pmc_note"], 80)
    cpu = out.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {"value": _num(cpu["value"]), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"), "sample": _short(cpu.get("sample", ""), 140)}
    if "identical_to_reference" in out:
        line["identical_to_reference"] = out["identical_to_reference"]
    if "speedup_vs_cpu_baseline_device_only" in out:
        line["speedup_vs_cpu_baseline_device_only"] = _num(out["speedup_vs_cpu_baseline_device_only"])
    if "bipartition_costs_per_s" in out:
        line["bipartition_costs_per_s"] = _num(out["bipartition_costs_per_s"])
    e2e = out.get("end_to_end")
    if e2e:
        line["end_to_end"] = {k: _num(e2e[k]) for k in ("value", "create_ms", "solve_and_getters_ms", "wall_ms", "fraction_of_device_only") if e2e.get(k) is not None}
    res = out.get("value_resident")
    if res:
        line["value_resident"] = {k: _num(res[k], 7 if k == "value" else 4) for k in ("value", "ms_per_step", "bipartition_costs_per_s") if res.get(k) is not None}
    if out.get("per_rank"):
        line["per_rank"] = [{k: _num(r.get(k), 3) for k in ("rank", "device", "tables", "create_ms", "solve_ms", "step_ms", "cpus", "numa_node") if r.get(k) is not None} for r in out["per_rank"][:8]]
    strict = out.get("value_8d_strict")
    if strict:
        line["value_8d_strict"] = {k: _num(v) if not isinstance(v, str) else _short(v, 120) for k, v in strict.items()}
    if "create_rate" in out:
        line["create_rate"] = {k: _num(v) if not isinstance(v, str) else _short(v, 100) for k, v in out["create_rate"].items()}
    entries = out.get("configs")
    if entries is not None:
        short = {}
        for c in entries:
            if "error" in c:
                short[c["name"]] = {"error": _short(c["error"], 60)}
                continue
            r = c.get("roofline") or {}
            rec = {"value": _num(c.get("value")), "res": _num((c.get("value_resident") or {}).get("value")), "ms": _num(c.get("ms_per_step")), "frac": _num(r.get("frac"), 3), "active": _num(r.get("valu_active_frac"), 3),
                   "us": _num(r.get("avg_launch_us"), 3), "cpu": _num((c.get("cpu_baseline") or {}).get("value"), 3), "ident": c.get("identical_to_reference"),
                   "e2e": _num((c.get("end_to_end") or {}).get("value")), "host8": _num((c.get("create_rate") or {}).get("host_over_8_devices"), 3),
                   "shim": _num(c.get("shim_over_native"), 3), "strict": _num((c.get("value_8d_strict") or {}).get("value"))}
            short[c["name"]] = {k: v for k, v in rec.items() if v is not None}
        line["configs"] = short
        line["configs_keys"] = "value columns/s, fresh tables (create inside the clock; genotype / heuristic / shim entries: as defined in the detail record); res columns/s with the tables resident; ms per step; frac VALU issue; active VALU busy; us per launch of the dominant kernel; cpu reference columns/s on 1 thread; ident == reference; e2e columns/s from host arrays; host8 fresh tables/s of this host / (8 x one device's tables/s); shim step time / native end to end; strict columns/s with whamd_dptable_create inside the clock (SURVEY 8d)"
    if detail_file:
        line["detail"] = detail_file
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:   # never again a line the driver cannot parse: drop the optional parts, largest first
        for key in ("configs_keys", "create_rate", "per_rank", "end_to_end", "value_8d_strict", "configs"):
            line.pop(key, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    if len(text) > LINE_LIMIT:   # still too long (a very long roofline / config string): the contract's head keys alone, whatever they hold
        head = {k: line[k] for k in HEAD_KEYS if k in line}
        head["config"] = {"workload": _short(cfg.get("workload", ""), 150)}
        if detail_file:
            head["detail"] = detail_file
        text = json.dumps(head, separators=(",", ":"))
    return text


def emit(out, args):
    """Full detail first (comment lines + a side file), the compact line LAST."""
    detail_file = None
    if not args.sub:
        entries = out.get("configs") or []
        head = {k: v for k, v in out.items() if k != "configs"}
        print("# bench detail headline: " + json.dumps(head), flush=True)
        for c in entries:
            print(f"# bench detail {c.get('name')}: " + json.dumps(c), flush=True)
        try:
            os.makedirs(os.path.dirname(args.detail_file), exist_ok=True)
            with open(args.detail_file, "w") as f:
                json.dump(out, f, indent=1)
            detail_file = os.path.relpath(args.detail_file, ROOT)
        except OSError:
            detail_file = None
        print(compact_line(out, detail_file), flush=True)
    else:
        print(json.dumps(out), flush=True)   # a child of the `configs` array: the parent reads the full record


def dominant_kernel(args, grouped=False):
    if args.path in ("column", "column_keys"):
        return "column_step_fused"
    if args.trio or args.quartet:
        return "resident_segment_ped" if args.path == "resident" else ("pedslot_group" if grouped else "pedslot_run")
    return "resident_segment" if args.path == "resident" else ("slot_group" if grouped else "slot_run")   # (substring: also slot_groupx / slot_runx, the X kernels)


def run_extra_configs(args):
    """The other single-GPU workloads, one child process each (own tables, own counters, own CPU sample)."""
    out = []
    for name in EXTRA_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--sub", "--steps", "6", "--warmup", "2",
               "--configs", "off", "--pmc", args.pmc, "--pmc-keep", os.path.join(args.pmc_keep, name)]
        if args.cpu_baseline_columns == 0:
            cmd += ["--cpu-baseline-columns", "0"]
        t0 = time.perf_counter()
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
            line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            if res.returncode != 0 or not line:
                out.append({"name": name, "error": f"rc={res.returncode} {res.stderr[-300:]}"})
                continue
            full = json.loads(line[-1])
        except Exception as exc:  # noqa: BLE001 -- the headline must come out whatever a child does
            out.append({"name": name, "error": repr(exc)})
            continue
        roof = full.get("roofline", {})
        entry = {
            "name": name,
            "workload": full["config"]["workload"],
            "value": full["value"],
            "unit": full["unit"],
            "ms_per_step": full["ms_per_step"],
            "ms_per_step_min": full.get("ms_per_step_min"),
            "ms_per_step_median": full.get("ms_per_step_median"),
            "steps": full["steps"],
            "warmup": full["warmup"],
            "tables_in_flight": full["config"].get("blocks_in_flight_per_gpu"),
            "tables_per_launch": full["config"].get("tables_per_launch"),
            "identical_to_reference": full.get("identical_to_reference"),
            "end_to_end": ({k: full["end_to_end"].get(k) for k in ("value", "fraction_of_device_only", "create_ms", "solve_and_getters_ms", "wall_ms", "tables_per_window", "create_threads", "host_threads_per_create", "tried")}
                           if "end_to_end" in full else None),
            "bipartition_costs_per_s": full["bipartition_costs_per_s"],
            "optimal_cost_checksum": full["config"]["optimal_cost_checksum"],
            "forward_launches_per_step": full["rank0"]["forward_launches_per_step"],
            "roofline": {k: roof.get(k) for k in ("bound", "kernel", "frac", "valu_active_frac", "work_bound_frac", "avg_launch_us", "peak", "unit", "pmc_note")},
            "wall_s": time.perf_counter() - t0,
        }
        if "create_rate" in full:
            entry["create_rate"] = full["create_rate"]
        for key in ("shim_over_native", "native_end_to_end", "step_pieces_ms", "value_resident", "value_8d_strict", "per_rank", "host_shapes_tried"):
            if key in full:
                entry[key] = full[key]
        if "cpu_baseline" in full:
            entry["cpu_baseline"] = {k: full["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample")}
            entry["speedup_vs_cpu_baseline_device_only"] = full["value"] / full["cpu_baseline"]["value"]
        out.append(entry)
    return out


def main():
    args = parse_args()
    if args.cpu_sample_worker:
        return cpu_sample_worker(args)
    if args.heuristic_cpu_worker:
        return heuristic_cpu_worker(args)
    if args.create_rate_worker:
        if args.workload:
            for key, value in WORKLOADS[args.workload].items():
                setattr(args, key, value)
        return create_rate_worker(args)
    explicit = any(a in sys.argv[1:] for a in ("--workload", "--trio", "--quartet", "--distrust", "--irregular", "--genotype", "--heuristic", "--shim", "--variants", "--coverage", "--blocks", "--blocks-per-gpu", "--path", "--option"))
    if args.workload:
        for key, value in WORKLOADS[args.workload].items():
            setattr(args, key, value)
    import __graft_entry__ as entry

    if not os.path.exists(os.path.join(ROOT, "whatshap_amd", "libwhatshap_amd.so")):
        entry.build()
    if args.genotype:
        return genotype_main(args)
    if args.heuristic:
        return heuristic_main(args)
    if args.shim:
        return shim_main(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from whatshap_amd import _native
    from whatshap_amd.blocks import assign_blocks, block_weight

    child = args.sub or args.pmc_inner   # children of this script: every table waits on its own stream, torch is not needed
    torch = None
    if not child:
        import torch  # device selection, synchronisation and the rendezvous only

    visible = _native.device_count()
    if visible < 1 or (torch is not None and not torch.cuda.is_available()):
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    device = local_rank % visible if args.oversubscribe else local_rank
    if device >= visible:
        raise SystemExit(f"rank {rank}: device {local_rank} requested, {visible} visible")
    if torch is not None:
        torch.cuda.set_device(device)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank, to exercise the path)
        import torch.distributed as dist

        # gloo: the data path has no collective (north_star: host-side work queue, no RCCL); the rendezvous only carries the
        # barrier, the max-over-ranks time and the per-rank checksums -- host tensors
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")

    blocks, scaling, workload = resolve_workload(args, world)
    T = 4 if args.trio else (16 if args.quartet else 1)
    weights = [block_weight(v, args.coverage, T) for _, v in blocks]
    mine = assign_blocks(weights, world)[rank]
    if world > 1 and not any(a.startswith("--in-flight") for a in sys.argv[1:]):
        # configs[4] over several GPUs: a rank keeps ALL its blocks in flight (at N = 2 / 4 that is 12 / 6 blocks sharing their launches, with
        # the layout for shared launches; at N = 8 three blocks on their own streams)
        args.in_flight = max(args.in_flight, len(mine))
        if len(mine) > 4 and not args.option:
            args.option = ["shared_launches=1"]
    # one process per GPU: this rank's host threads (creates, result extraction) stay on the CPUs next to ITS GPU (whatshap_amd.blocks.bind_rank_to_device_cpus)
    cpu_binding = None
    if not args.no_affinity:
        from whatshap_amd.blocks import bind_rank_to_device_cpus

        if world > 1 or "RANK" in os.environ:
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
            cpu_binding = bind_rank_to_device_cpus(local_rank, local_world, devices=[(r % visible if args.oversubscribe else r) for r in range(local_world)])
        else:
            # one rank: the whole NUMA node of its GPU, not the whole host -- unbound, Python's worker threads (the creates of blocks.solve_blocks) land on
            # both sockets while the library's own workers stay on one: 24 creates took 55 ms on 256 unbound CPUs and 31 ms in a process held to 32
            cpu_binding = bind_rank_to_device_cpus(0, 1, devices=[device])
    problems = [build_block(args, *blocks[b]) for b in mine]
    native_path = None if args.path == "auto" else args.path
    tables = []
    for problem in problems:
        t = _native.NativeTable(problem, device=device, path=native_path, solve=False, options=option_dict(args))
        tables.append(t)

    def step():
        # host-side work queue: `in_flight` blocks of this rank at a time on their own streams (launch sequences
        # interleaved, so that they start together), then collected
        for start in range(0, len(tables), args.in_flight):
            window = tables[start:start + args.in_flight]
            _native.enqueue_many(window)
            _native.wait_many(window)

    def sync():
        if torch is not None:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if torch is not None:
            torch.cuda.synchronize()

    # ---------------------------------------------------------------- region 1: tables RESIDENT (created before the clock) -> value_resident, roofline
    for _ in range(args.warmup):
        step()
    if args.pmc_inner:
        step()
        return
    sync()
    t0 = time.perf_counter()
    fwd_ms = bt_ms = 0.0
    launches = 0
    step_s = []
    grouped = False
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_s.append(time.perf_counter() - ts)
        for start in range(0, len(tables), args.in_flight):
            st = [t.stats() for t in tables[start:start + args.in_flight]]
            if st[0]["group_tables"] > 1:
                # the tables of the window shared their launches (enqueue_many -> one slot_group launch per super-step): every table
                # reports the group's forward time and the launches it took part in
                grouped = True
                fwd_ms += max(x["forward_ms"] for x in st)
                launches += max(x["forward_launches"] for x in st)
            else:
                fwd_ms += sum(x["forward_ms"] for x in st)
                launches += sum(x["forward_launches"] for x in st)
            bt_ms += sum(x["backtrace_ms"] for x in st)
    sync()
    elapsed_resident = time.perf_counter() - t0
    resident_step_s = list(step_s)
    stats = [t.stats() for t in tables]
    totals = [float(sum(s["n_columns"] for s in stats)), float(sum(s["n_costs"] for s in stats)), float(sum(t.optimal_score() for t in tables))]
    for t in tables:   # the fresh tables below need the room (arena blocks go back to the cache)
        t.release_device()

    # ---------------------------------------------------------------- region 2: FRESH tables, SURVEY.md 8(d)'s clock: "from entering the constructor" --
    # every step creates its tables from the flattened host arrays (whamd_dptable_create: columns, indexing schemes, cost terms, plan, upload), solves them
    # (forward, backtrace, path download, superread assembly) and destroys them.  Several tables: the host-side work queue (blocks.solve_blocks) creates the
    # next window under the device solve of the current one.  This is `value`.
    from whatshap_amd.blocks import close_tables, solve_blocks

    n_cpus = len(os.sched_getaffinity(0))
    from whatshap_amd.blocks import cpu_quota, host_cpu_budget
    quota = cpu_quota()
    budget = host_cpu_budget(n_cpus, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) if world > 1 else 1, quota)   # threads this rank can keep busy: its CPU slice, capped by its share of the control group's quota
    whole = min(args.in_flight, len(problems))
    if len(problems) == 1:
        host_shapes = [(1, 0, 1, 1)]
    elif world > 1:
        # a rank's CPU slice (32 hardware threads of the box at N = 8) over its tables: few large tables (configs[4] at N = 8: three per rank) get many threads each
        workers = max(1, min(len(problems), n_cpus))
        host_shapes = [(workers, max(1, min(32, n_cpus // workers)), whole), (max(1, min(len(problems), n_cpus // 2)), 2, whole), (max(1, min(len(problems), n_cpus // 4)), 4, whole)]
    else:
        # (workers, threads per create, tables per window): one window of everything keeps the device's launch sequence shortest; two or three windows let the
        # creates of the next one run under the solve of the current one -- what wins depends on the table shape and is measured, not assumed.  The busy threads of
        # the default shape are the rank's CPU budget: more than the control group's quota and the whole process is frozen for the rest of every period
        # (blocks.cpu_quota: 16 CPUs on the boxes this was measured on -- 16 workers x 2 threads, the default until then, was twice the budget)
        # default: a burst (16 workers x 2 threads); one single-threaded worker per CPU of the control group's quota (paced: never frozen) is tried beside it
        w1 = max(1, min(len(problems), budget))
        burst, paced = (max(1, min(16, len(problems))), 2, whole), (w1, 1, whole)
        host_shapes = [burst, paced]
        host_shapes += [(max(1, min(len(problems), 32)), 1, whole)]
        if whole >= 24:
            # ... two or three windows, and two windows on the device at once: window k + 1 is enqueued -- its own stream -- before window k is collected
            first = host_shapes[0]
            host_shapes += [(first[0], first[1], (whole + 1) // 2), (first[0], first[1], (whole + 2) // 3), (first[0], first[1], (whole + 1) // 2, 2)]
    host_shapes = list(dict.fromkeys(tuple(h) + (1,) * (4 - len(h)) for h in host_shapes))   # (workers, threads per create, tables per window, windows on the device)

    def fresh_step(shape):
        ta = time.perf_counter()
        if len(problems) == 1:
            t = _native.NativeTable(problems[0], device=device, path=native_path, solve=False, options=option_dict(args))
            tb = time.perf_counter()
            _native.enqueue_many([t])
            _native.wait_many([t])
            tc = time.perf_counter()
            st = t.stats()
            checksum = int(t.optimal_score())
            t.close()
            td = time.perf_counter()
            return {"wall_ms": (td - ta) * 1e3, "create_ms": (tb - ta) * 1e3, "solve_ms": (tc - tb) * 1e3, "close_ms": (td - tc) * 1e3,
                    "device_ms": st["total_ms"], "superreads_ms": st["host_finish_ms"], "flatten_ms": st.get("host_flatten_ms"), "checksum": checksum}
        trace = []
        solved = solve_blocks(problems, device=device, path=native_path, max_in_flight=shape[2], release=True,
                              create_threads=shape[0], host_threads_per_create=shape[1], windows_on_device=shape[3], trace=trace)
        tc = time.perf_counter()
        checksum = int(sum(t.optimal_score() for t in solved))
        close_tables(solved)     # (on a few threads: 7 ms for 96 tables one after the other)
        td = time.perf_counter()
        # the submitting thread's time line: waiting for a window's creates (not hidden under a solve) / enqueue + device + result extraction
        create_wait = solve_ms = 0.0
        last = 0.0
        for what, _wi, ms in trace:
            if what == "created":
                create_wait += ms - last
            elif what == "collected":
                solve_ms += ms - last
            elif what == "enqueued":
                solve_ms += ms - last
            last = ms
        return {"wall_ms": (td - ta) * 1e3, "create_ms": create_wait, "solve_ms": solve_ms, "close_ms": (td - tc) * 1e3, "checksum": checksum,
                "other_ms": (tc - ta) * 1e3 - create_wait - solve_ms}   # (the work queue's own tail: device releases on the pool's threads, the pool's shutdown)

    tried = []
    shape = host_shapes[0]
    fresh_step(shape)              # untimed: the first fresh step of the process also sizes the host and device pools
    for cand in host_shapes:       # untimed: two steps per host shape (one step alone is too noisy to choose by)
        walls = [fresh_step(cand)["wall_ms"] for _ in range(2 if len(host_shapes) > 1 else 1)]
        tried.append({"create_threads": cand[0], "host_threads_per_create": cand[1], "tables_per_window": cand[2], "windows_on_device": cand[3], "wall_ms": min(walls), "wall_ms_slower": max(walls)})
    if len(host_shapes) > 1:
        # the default shape (16 workers x 2 threads, one window of everything) stays unless another one beat it by 7 % in BOTH of its steps: a shape picked on one
        # lucky step cost up to 15 % of the timed region
        best = tried[0]
        for r in tried[1:]:
            if r["wall_ms_slower"] < 0.93 * best["wall_ms"]:
                best = r
        shape = (best["create_threads"], best["host_threads_per_create"], best["tables_per_window"], best["windows_on_device"])
    for _ in range(max(0, args.warmup - 1 - len(host_shapes))):
        fresh_step(shape)
    sync()
    t0 = time.perf_counter()
    fresh = []
    for _ in range(args.steps):
        fresh.append(fresh_step(shape))
    sync()
    elapsed = time.perf_counter() - t0
    step_s = [r["wall_ms"] * 1e-3 for r in fresh]
    if any(r["checksum"] != int(totals[2]) for r in fresh):
        raise SystemExit(f"rank {rank}: a fresh solve's cost checksum {[r['checksum'] for r in fresh]} differs from the resident tables' {int(totals[2])}")

    def med(key):
        vals = sorted(r[key] for r in fresh if r.get(key) is not None)
        return vals[len(vals) // 2] if vals else None

    per_rank = {"rank": rank, "device": device, "tables": len(problems), "create_ms": med("create_ms"), "solve_ms": med("solve_ms"), "close_ms": med("close_ms"), "other_ms": med("other_ms"),
                "step_ms": med("wall_ms"), "resident_step_ms": sorted(resident_step_s)[len(resident_step_s) // 2] * 1e3,
                "cpus": (cpu_binding or {}).get("n_cpus", n_cpus), "cpu_quota": quota, "cpu_budget": budget, "numa_node": (cpu_binding or {}).get("node"), "cpu_source": (cpu_binding or {}).get("source", "unbound"),
                "create_threads": shape[0], "host_threads_per_create": shape[1], "tables_per_window": shape[2], "windows_on_device": shape[3]}
    per_rank_checksums = [int(totals[2])]
    # what a SCALE record can be audited with: which rank ran which blocks on which device, and for how long
    print(f"[bench rank {rank}/{world}] device {device}, blocks {[blocks[b][0] for b in mine]} (seeds), {len(mine)} table(s), fresh: {elapsed:.3f} s for {args.steps} step(s) "
          f"(create {per_rank['create_ms']:.1f} ms + solve {per_rank['solve_ms']:.1f} ms + close {per_rank['close_ms']:.1f} ms per step), resident: {elapsed_resident:.3f} s, "
          f"cpus {per_rank['cpus']} ({per_rank['cpu_source']}, node {per_rank['numa_node']}), cost checksum {int(totals[2])}", file=sys.stderr, flush=True)
    per_rank_all = [per_rank]
    if dist is not None:
        import torch as _torch

        tmax = _torch.tensor([elapsed, elapsed_resident], dtype=_torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, elapsed_resident = float(tmax[0].item()), float(tmax[1].item())
        gathered = [None] * world
        dist.all_gather_object(gathered, totals + [float(device), per_rank])
        per_rank_checksums = [int(g[2]) for g in gathered]
        per_rank_devices = [int(g[3]) for g in gathered]
        per_rank_all = [g[4] for g in gathered]
        totals = [sum(g[i] for g in gathered) for i in range(3)]
    else:
        per_rank_devices = [device]
    cols_job, costs_job = totals[0], totals[1]

    if rank == 0:
        avg_launch_us = fwd_ms * 1e3 / max(launches, 1)
        bytes_rank = sum(s["algorithmic_bytes"] for s in stats)
        bytes_per_launch = bytes_rank / max(launches / args.steps, 1)
        column_path = args.path in ("column", "column_keys")
        kernel = dominant_kernel(args, grouped)
        kind = ("synthetic trio PedMEC, genotypes not trusted" if args.distrust else "synthetic trio PedMEC") if args.trio else (("synthetic quartet PedMEC (two trios sharing parents), genotypes not trusted" if args.distrust else "synthetic quartet PedMEC (two trios sharing parents)") if args.quartet else "synthetic diploid single-individual")
        out = {
            "metric": "variant-columns/sec at max-coverage %d (bipartition-costs/sec reported alongside)" % args.coverage,
            "value": cols_job * args.steps / elapsed,
            "unit": "variant-columns/s",
            "bipartition_costs_per_s": costs_job * args.steps / elapsed,
            "bipartition_costs_note": ("sum over columns of 2^k_c * T (SURVEY.md 8d metric 2), credited in full: for a single individual the "
                                       "kernels EVALUATE only half of them, the other half follows from the complement symmetry D[~x] = D[x]")
                                      if T == 1 else "sum over columns of 2^k_c * T (SURVEY.md 8d metric 2); every one is evaluated",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_min": min(step_s) * 1e3,
            "ms_per_step_median": sorted(step_s)[len(step_s) // 2] * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": kind + f", max-coverage {args.coverage}: " + workload,
                "blocks": len(blocks),
                "blocks_per_rank": [len(r) for r in assign_blocks(weights, world)],
                "block_seeds_per_rank": [[blocks[b][0] for b in r] for r in assign_blocks(weights, world)],
                "device_per_rank": per_rank_devices,
                "blocks_in_flight_per_gpu": min(args.in_flight, len(mine)),
                "tables_per_launch": (min(args.in_flight, len(mine)) if grouped else 1),
                "max_coverage": args.coverage,
                "transmission_values": T,
                "path": args.path,
                "options": args.option,
                "optimal_cost_checksum": int(totals[2]),
                "optimal_cost_checksum_per_rank": per_rank_checksums,
                "rendezvous": "gloo (barrier, max-over-ranks time, checksums; no collective on the data path)" if dist is not None else "none",
            },
            "rank0": {"forward_ms_per_step": fwd_ms / args.steps, "backtrace_ms_per_step": bt_ms / args.steps,
                      "forward_launches_per_step": launches / args.steps},
            "value_is": "FRESH tables: every timed step runs whamd_dptable_create (columns, indexing schemes, cost terms, plan, upload) from the flattened host arrays, the "
                        "solve (forward, backtrace, path download, superread assembly) and the destroy of every table of the rank -- SURVEY.md 8(d)'s clock starts at the "
                        "constructor; several tables go through the host-side work queue (next window created under the current solve)",
            "value_resident": {"value": cols_job * args.steps / elapsed_resident, "ms_per_step": elapsed_resident / args.steps * 1e3,
                               "ms_per_step_min": min(resident_step_s) * 1e3, "ms_per_step_median": sorted(resident_step_s)[len(resident_step_s) // 2] * 1e3,
                               "bipartition_costs_per_s": costs_job * args.steps / elapsed_resident,
                               "what": "the same steps on tables created BEFORE the clock (plan and operand tables resident in HBM): forward + backtrace + result extraction only -- what "
                                       "rounds 1-5 reported as `value`; the roofline's launch time is measured here"},
            "per_rank": per_rank_all,
            "host_shapes_tried": tried,
        }
        out["config"]["cpu_binding"] = (f"rank r bound to the CPUs of its GPU's NUMA node, shared evenly between the ranks of that node (whatshap_amd.blocks.bind_rank_to_device_cpus): "
                                        f"{(cpu_binding or {}).get('n_cpus')} CPUs, {(cpu_binding or {}).get('source')}") if cpu_binding is not None else "none (--no-affinity)"
        # ---- fresh tables end to end (host-inclusive): create + solve + getters
        if world == 1 and len(blocks) == 1:
            seed, v = blocks[mine[0]]
            problem = build_block(args, seed, v)

            def fresh_table():
                te0 = time.perf_counter()
                fresh = _native.NativeTable(problem, device=device, path=None if args.path == "auto" else args.path, solve=False)
                apply_options(fresh, args)
                te1 = time.perf_counter()
                fresh.solve()
                fresh.optimal_score(), fresh.super_reads(), fresh.partitioning()
                te2 = time.perf_counter()
                fresh_stats.update(fresh.stats())
                fresh.close()
                return (te1 - te0) * 1e3, (te2 - te1) * 1e3

            fresh_stats = {}
            fresh_table()   # (the first create of a process also sizes the pinned staging area)
            create_ms, rest_ms = fresh_table()
            # "from entering the constructor to index_path complete, excluding ReadSet flattening and superread assembly".  The
            # constructor of the reference (src/pedigreedptable.cpp:15-37) walks the columns, builds the indexing schemes and runs compute_table;
            # here that is ALL of whamd_dptable_create (columns + indexing scheme + cost terms, plan, upload: its input is the flattened CSR view)
            # plus the device's forward pass, backtrace and path download; whamd_dptable_wait's superread assembly (host_finish_ms) stays outside.
            strict_ms = create_ms + fresh_stats["total_ms"]
            out["value_8d_strict"] = {"value": v / (strict_ms * 1e-3), "unit": "variant-columns/s", "ms": strict_ms, "create_ms": create_ms,
                                      "flatten_ms": fresh_stats.get("host_flatten_ms"),
                                      "terms_plan_upload_ms": (create_ms - fresh_stats["host_flatten_ms"]) if fresh_stats.get("host_flatten_ms") is not None else None,
                                      "device_ms": fresh_stats["total_ms"], "superreads_ms_excluded": fresh_stats["host_finish_ms"],
                                      "what": "SURVEY 8(d) to the letter for ONE fresh table: whamd_dptable_create (flatten_ms = ColumnIterator's columns; the rest = indexing scheme, cost terms, plan, upload) + forward + backtrace + path download; superread assembly and destroy excluded (`value` includes both)"}
            out["end_to_end"] = {"value": v / ((create_ms + rest_ms) * 1e-3), "unit": "variant-columns/s", "create_ms": create_ms,
                                 "solve_and_getters_ms": rest_ms, "host_threads": min(os.cpu_count() or 1, 32),
                                 "fraction_of_device_only": (v / ((create_ms + rest_ms) * 1e-3)) / out["value_resident"]["value"],
                                 "what": "whamd_dptable_create (flatten + plan + upload) + solve + 3 getters of ONE fresh table from host arrays"}
            if not args.sub:
                # the same with the create path held to 8 host threads (WHAMD_PLAN_THREADS; the default is min(hardware threads, 32))
                saved = os.environ.get("WHAMD_PLAN_THREADS")
                os.environ["WHAMD_PLAN_THREADS"] = "8"
                try:
                    create8, rest8 = fresh_table()
                finally:
                    if saved is None:
                        del os.environ["WHAMD_PLAN_THREADS"]
                    else:
                        os.environ["WHAMD_PLAN_THREADS"] = saved
                out["end_to_end"]["create_ms_8_threads"] = create8
                out["end_to_end"]["value_8_threads"] = v / ((create8 + rest8) * 1e-3)
        elif world == 1:
            # several tables: `value` already is the host-side work queue from host arrays (region 2); end to end adds the three getters of every table
            te0 = time.perf_counter()
            solved = solve_blocks(problems, device=device, path=native_path, max_in_flight=shape[2], release=True, create_threads=shape[0], host_threads_per_create=shape[1], windows_on_device=shape[3])
            checksum = 0
            for t in solved:
                checksum += t.optimal_score()
                t.super_reads(), t.partitioning()
            wall = time.perf_counter() - te0
            for t in solved:
                t.close()
            if checksum != int(totals[2]):
                raise SystemExit(f"end_to_end: cost checksum {checksum} of the pipelined solve differs from {int(totals[2])}")
            out["end_to_end"] = {"value": cols_job / wall, "unit": "variant-columns/s", "wall_ms": wall * 1e3, "tables_per_window": shape[2], "create_threads": shape[0],
                                 "host_threads_per_create": shape[1], "fraction_of_device_only": (cols_job / wall) / out["value_resident"]["value"], "tried": tried,
                                 "what": f"{len(problems)} fresh tables from host arrays through blocks.solve_blocks (create of the next window on `create_threads` host workers x "
                                         f"`host_threads_per_create` threads under the device solve of the current one, enqueue_many / wait_many per window) + 3 getters per table"}
            # ---- how many fresh tables per second ONE RANK'S SHARE of this host hands to its device (one node's host feeds eight GPUs): creates only, in a child process
            # bound to the CPU slice rank 0 of 8 would get (bind_rank_to_device_cpus), against the rate at which one device solves such tables with its tables resident
            device_tables_per_s = len(problems) * args.steps / elapsed_resident
            rate = {"device_tables_per_s": device_tables_per_s, "host_nproc": os.cpu_count()}
            try:
                _native.release_caches()   # (the child sizes its arenas by the device's FREE memory: this process's kept arenas and pools go back first -- eight trio tables are 127 GB)
                cmd = [sys.executable, os.path.abspath(__file__), "--create-rate-worker", "0/8", "--coverage", str(args.coverage), "--variants", str(blocks[0][1]),
                       "--blocks", str(len(problems)), "--path", args.path] + workload_flags(args)
                for kv in args.option:
                    cmd += ["--option", kv]
                res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
                if res.returncode == 0 and line:
                    rate.update(json.loads(line[-1]))
                    rate["host_over_8_devices"] = rate["tables_per_s"] / device_tables_per_s
                    rate["what"] = (f"{len(problems)} whamd_dptable_create calls (flatten + plan + upload, no solve) in a process held to the CPUs rank 0 of 8 gets "
                                    f"({rate.get('cpus')} of {os.cpu_count()}: the NUMA node of its GPU shared evenly), best of the host shapes tried; against the tables per second ONE device "
                                    f"solves with its tables resident: below 1.0 an 8-GPU node is bound by its host.  (Rounds 4-5 divided the WHOLE host's single-process rate by 8.)")
                else:
                    rate["error"] = f"rc={res.returncode} {res.stderr[-200:]}"
            except Exception as exc:  # noqa: BLE001 -- never costs the line
                rate["error"] = repr(exc)
            out["create_rate"] = rate
        # ---- counters of the dominant kernel
        pmc, pmc_note = None, "skipped"
        want_pmc = args.pmc == "on" or (args.pmc == "auto" and world == 1 and not column_path)
        if want_pmc:
            for t in tables:
                t.release_device()
            try:
                pmc, pmc_note = run_pmc_passes(args, kernel, args.pmc_keep)
            except Exception as exc:  # noqa: BLE001 -- the bench line must come out whatever the profiler does
                pmc_note = repr(exc)
        out["roofline"] = roofline_from_counters(pmc, avg_launch_us, kernel, bytes_per_launch, args)
        out["roofline"]["pmc_note"] = pmc_note
        # work-based bound: the fewest VALU lane-operations the algorithm needs per evaluated (cell, transmission value) of a
        # column -- single individual: A, K - A, min3, accumulate + ending reads = 5, half of the cells by symmetry; trio (NF = 2): 2 adds + min,
        # 2 x 6 butterfly, accumulate, ending reads = 19; quartet 31 -- issued at the chip's 512 wave-instructions per cycle, over the forward time
        ops = {1: 5.0, 4: 19.0, 16: 31.0}[T]
        if T == 1 and args.path == "auto" and not args.distrust:
            # Y-form runs: one absolute-difference-accumulate per evaluated cell-column, a quarter of the add that forms the thread's
            # operand, and per ending read (0.5 per column) compare, shift-in and maximum on each pair: 1 + 0.25 + 1.5 = 2.75
            ops = 2.75
        evaluated = costs_job / world * (0.5 if T == 1 else 1.0)
        out["roofline"]["work_bound_frac"] = (evaluated * ops / 64.0 / (N_SIMD / 2.0) / (CLOCK_GHZ * 1e9)) / max(fwd_ms / args.steps * 1e-3, 1e-12)
        out["roofline"]["work_bound_note"] = f"{ops:g} VALU lane-operations per evaluated cell-value of a column at 512 wave-instructions/cycle, over the forward time of a step"
        # ---- CPU baseline (rank 0, N = 1 only)
        args.variants_for_cpu = blocks[0][1]
        args.device_for_parity = device
        if world == 1 and args.cpu_baseline_columns != 0:
            out["cpu_baseline"] = cpu_baseline(args, blocks[0][0])
            out["identical_to_reference"] = out["cpu_baseline"].pop("identical_to_reference")
            out["identical_what"] = out["cpu_baseline"].pop("identical_what")
            out["speedup_vs_cpu_baseline_device_only"] = out["value_resident"]["value"] / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        procs = args.cpu_baseline_procs if args.cpu_baseline_procs >= 0 else (os.cpu_count() or 1)
        if world == 1 and procs > 0:
            out["cpu_baseline_all_cores"] = cpu_baseline_procs(args, procs)
        # ---- the other single-GPU workloads of BASELINE.json (+ an irregular layout, + a quartet)
        if args.configs == "on" or (args.configs == "auto" and world == 1 and not explicit and not args.sub):
            for t in tables:
                t.release_device()
            out["configs"] = run_extra_configs(args)
        emit(out, args)
        parity = [out.get("identical_to_reference")] + [c.get("identical_to_reference") for c in out.get("configs", [])]
        if any(x is False for x in parity):
            print("bench.py: a device solution differs from the reference's (identical_to_reference: false)", file=sys.stderr, flush=True)
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
