"""Where the wall time of many fresh tables goes: create (1 / 4 / 8 / 16 host threads), enqueue_many + wait_many, getters."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block
n, cols, cov = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
problems = [synthetic_block(cols, cov, seed=100 + i) for i in range(n)]
for rep in range(2):
    for threads in [int(x) for x in os.environ.get("WHAMD_E2E_THREADS", "1,4,8,16").split(",")]:
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as pool:
            tables = list(pool.map(lambda p: _native.NativeTable(p, solve=False), problems))
        t1 = time.perf_counter()
        _native.enqueue_many(tables)
        _native.wait_many(tables)
        t2 = time.perf_counter()
        for t in tables:
            t.optimal_score(); t.super_reads(); t.partitioning()
        t3 = time.perf_counter()
        for t in tables:
            t.close()
        t4 = time.perf_counter()
        print(f"rep {rep} create with {threads:2d} threads {1e3 * (t1 - t0):7.1f} ms | solve {1e3 * (t2 - t1):7.1f} | getters {1e3 * (t3 - t2):6.1f} | close {1e3 * (t4 - t3):6.1f}", flush=True)
