"""Creates only (no solve): tables per second for several (workers x threads per create) shapes.  With WHAMD_USE_DEBUG_LIB=1 the debug switches apply
(WHAMD_SKIP_SLAB_COPY=1: the staging image is built but not sent -- results invalid, the rate without the link).  Usage: gpu_create_rate_ab.py [tables columns coverage]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from whatshap_amd import _native
from whatshap_amd.blocks import bind_rank_to_device_cpus
from whatshap_amd.synthetic import synthetic_block
if os.environ.get("WHAMD_USE_DEBUG_LIB"):
    _native.use_debug_library()
k, n, cov = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (96, 50000, 15)
if not os.environ.get("WHAMD_NO_BIND"):   # (WHAMD_NO_BIND=1: both sockets -- twice the memory channels)
    bind_rank_to_device_cpus(0, 1, devices=[0])
problems = [synthetic_block(n, cov, seed=100 + i) for i in range(k)]
shapes = ((16, 2), (32, 2), (32, 1), (64, 1), (96, 1), (8, 4))
if os.environ.get("WHAMD_RATE_SHAPES"):   # e.g. "32x1,16x2"
    shapes = tuple(tuple(int(v) for v in item.split("x")) for item in os.environ["WHAMD_RATE_SHAPES"].split(","))
for workers, per in shapes:
    opts = {"shared_launches": "1", "host_threads": str(per)}
    walls = []
    for rep in range(4):
        with ThreadPoolExecutor(max_workers=workers) as pool:
            t0 = time.perf_counter()
            made = list(pool.map(lambda pr: _native.NativeTable(pr, solve=False, options=opts), problems))
            walls.append(time.perf_counter() - t0)
        for t in made:
            t.close()
    best = min(walls[1:])
    print(f"{workers:3d} workers x {per} threads: {k} creates in {best * 1e3:6.1f} ms = {k / best:5.0f} tables/s  (all: {' '.join(f'{w * 1e3:.0f}' for w in walls)})", flush=True)
