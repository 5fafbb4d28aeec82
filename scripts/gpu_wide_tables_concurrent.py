"""Three full-width tables (coverage 20) solved at once on their own streams, resident: step time with the uploads on the shared high-priority upload streams (product)
and, with WHAMD_USE_DEBUG_LIB=1 WHAMD_UPLOAD_ON_TABLE_STREAM=1, on the tables' own streams (no priority stream is ever created).  Usage: gpu_wide_tables_concurrent.py [tables columns coverage]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
from whatshap_amd.blocks import bind_rank_to_device_cpus
from whatshap_amd.synthetic import synthetic_block
if os.environ.get("WHAMD_USE_DEBUG_LIB"):
    _native.use_debug_library()
k, n, cov = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 100000, 20)
bind_rank_to_device_cpus(0, 1, devices=[0])
tables = [_native.NativeTable(synthetic_block(n, cov, seed=100 + i), solve=False) for i in range(k)]
for rep in range(5):
    t0 = time.perf_counter()
    _native.enqueue_many(tables)
    _native.wait_many(tables)
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {k} tables x {n} columns at coverage {cov}: {dt * 1e3:.1f} ms = {k * n / dt / 1e6:.2f} M columns/s; forward of table 0 {tables[0].stats()['forward_ms']:.1f} ms", flush=True)
t0 = time.perf_counter(); tables[0].solve(); print(f"one table alone: {(time.perf_counter() - t0) * 1e3:.1f} ms")
