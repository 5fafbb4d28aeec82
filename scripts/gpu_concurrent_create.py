"""Phases of whamd_dptable_create when MANY tables are created at once (WHAMD_DEBUG_TIMING lines averaged): where a create spends its wall time under
concurrency.  Usage: gpu_concurrent_create.py tables workers threads_per_create [columns coverage]"""
import os, re, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] != "--inner":
    out = subprocess.run([sys.executable, __file__, "--inner"] + sys.argv[1:], capture_output=True, text=True, env=dict(os.environ, WHAMD_DEBUG_TIMING="1"))
    print(out.stdout, end="")
    sums, cnt = {}, {}
    for line in out.stderr.splitlines():
        if line.strip() == "rep 1":
            sums, cnt = {}, {}     # (the first round sizes the pools)
        for m in re.finditer(r"(flatten|plan \+ upload|plan|descriptors \+ copies|rest|column ranges|layouts|concatenation) ([0-9.]+) ms", line):
            key = ("create: " if line.startswith("[whamd timing] create") else ("upload: " if "upload:" in line else "slot plan: ")) + m.group(1)
            sums[key] = sums.get(key, 0.0) + float(m.group(2)); cnt[key] = cnt.get(key, 0) + 1
        m = re.search(r"upload: ([a-z ,:+()/]+?) ([0-9.]+) ms$", line)
        if m and line.startswith("[whamd timing]   upload"):
            key = "  " + m.group(1); sums[key] = sums.get(key, 0.0) + float(m.group(2)); cnt[key] = cnt.get(key, 0) + 1
        m = re.search(r"flatten: ([a-z ,:+()/]+?) ([0-9.]+) ms$", line)
        if m:
            key = "  flatten " + m.group(1); sums[key] = sums.get(key, 0.0) + float(m.group(2)); cnt[key] = cnt.get(key, 0) + 1
    for key in sums:
        print(f"{key:70s} mean {sums[key] / cnt[key]:7.2f} ms over {cnt[key]} creates")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from whatshap_amd import _native
from whatshap_amd.blocks import bind_rank_to_device_cpus
from whatshap_amd.synthetic import synthetic_block
k, workers, per = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n, cov = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (50000, 15)
bind_rank_to_device_cpus(0, 1, devices=[0])
problems = [synthetic_block(n, cov, seed=100 + i) for i in range(k)]
opts = {"shared_launches": "1", "host_threads": str(per)}
for rep in range(3):
    print(f"rep {rep}", file=sys.stderr, flush=True)
    with ThreadPoolExecutor(max_workers=workers) as pool:
        t0 = time.perf_counter()
        made = list(pool.map(lambda pr: _native.NativeTable(pr, solve=False, options=opts), problems))
        wall = time.perf_counter() - t0
    for t in made:
        t.close()
    print(f"rep {rep}: {k} creates on {workers} x {per} threads: {wall * 1e3:.1f} ms = {k / wall:.0f} tables/s", flush=True)
