#!/usr/bin/env python3
"""Counter probe of one bench workload: rocprofv3 --kernel-trace --pmc passes (never combined with other trace domains) on
`bench.py --pmc-inner --workload W`, per-dispatch averages over the widest dispatches of the named kernel.
usage: gpu_pmc_probe.py WORKLOAD KERNEL_SUBSTRING [--variants N] [--env K=V ...] -- "COUNTERS OF PASS 1" "COUNTERS OF PASS 2" ..."""
import collections, csv, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
split = args.index("--")
head, passes = args[:split], args[split + 1:]
workload, kernel = head[0], head[1]
variants = head[head.index("--variants") + 1] if "--variants" in head else "8000"
extra_env = {}
for i, a in enumerate(head):
    if a == "--env":
        k, _, v = head[i + 1].partition("=")
        extra_env[k] = v
env = dict(os.environ, TMPDIR="/tmp", **extra_env)
for counters in passes:
    out_dir = tempfile.mkdtemp(prefix="whamd_probe_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", out_dir, "-o", "p", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--pmc-inner", "--workload", workload, "--variants", variants, "--steps", "1", "--warmup", "1", "--configs", "off"]
    res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    files = [os.path.join(b, n) for b, _, ns in os.walk(out_dir) for n in ns if n.endswith("counter_collection.csv")]
    if res.returncode != 0 or not files:
        print("pass failed:", counters, res.returncode, res.stderr[-300:])
        continue
    rows = [r for r in csv.DictReader(open(files[0])) if kernel in r["Kernel_Name"]]
    if not rows:
        print("kernel not in trace:", kernel)
        continue
    full = max(int(r["Grid_Size"]) for r in rows)
    sums, cnt = collections.Counter(), collections.Counter()
    for r in rows:
        if int(r["Grid_Size"]) == full:
            sums[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
    print(f"# {workload} {kernel}: grid {full}, {max(cnt.values())} dispatches")
    for c in counters.split():
        if cnt[c]:
            print(f"{c:28s} {sums[c] / cnt[c]:16.1f}")
