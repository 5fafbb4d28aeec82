"""Host only (no device call): whamd_plan_summarize -- flatten + plan of a table, what a create does before it touches the device -- from 1 .. 96 threads at once,
one thread per table (WHAMD_PLAN_THREADS=1).  Thread-ms per table should stay flat if the host part of a create scaled with the cores.
Usage: host_plan_scaling.py [tables columns coverage]"""
import os, sys, time
os.environ.setdefault("WHAMD_PLAN_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block
k, n, cov = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (96, 50000, 15)
if not os.environ.get("WHAMD_NO_BIND"):
    try:
        from whatshap_amd.blocks import bind_rank_to_device_cpus
        bind_rank_to_device_cpus(0, 1, devices=[0])
    except Exception as e:  # noqa: BLE001  (no device: the whole machine)
        print("not bound:", e)
problems = [synthetic_block(n, cov, seed=100 + i) for i in range(k)]
_native.plan_summary(problems[0])
for workers in (1, 8, 16, 32, 64, 96):
    best = 1e9
    for rep in range(3):
        with ThreadPoolExecutor(max_workers=workers) as pool:
            t0 = time.perf_counter()
            list(pool.map(_native.plan_summary, problems))
            best = min(best, time.perf_counter() - t0)
    print(f"{workers:3d} threads: {k} tables in {best * 1e3:7.1f} ms = {k / best:6.0f} tables/s, {best * 1e3 / k * min(workers, k):5.1f} thread-ms per table", flush=True)
