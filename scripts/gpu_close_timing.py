"""Where a fresh step's `close` goes: python-level wall of NativeTable.close() after enqueue_many / wait_many, next to the library's own split
(WHAMD_DEBUG_TIMING: destroy device side / host side).  Usage: gpu_close_timing.py columns coverage tables [trio]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WHAMD_DEBUG_TIMING"] = "1"
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block

n, cov, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
trio = len(sys.argv) > 4 and sys.argv[4] == "trio"
problems = [synthetic_block(n, cov, seed=100 + i, trio=trio) for i in range(k)]
opts = {"shared_launches": "1"} if k > 4 else None
for rep in range(3):
    t0 = time.perf_counter()
    tables = [_native.NativeTable(p, solve=False, options=opts) for p in problems]
    t1 = time.perf_counter()
    _native.enqueue_many(tables); _native.wait_many(tables)
    t2 = time.perf_counter()
    for t in tables: t.release_device()
    t3 = time.perf_counter()
    sys.stderr.flush()
    marks = []
    for t in tables:
        a = time.perf_counter(); t.close(); marks.append((time.perf_counter() - a) * 1e3)
    t4 = time.perf_counter()
    print(f"rep {rep}: create {1e3 * (t1 - t0):.1f} ms, solve {1e3 * (t2 - t1):.1f} ms, release_device {1e3 * (t3 - t2):.1f} ms, close {1e3 * (t4 - t3):.1f} ms "
          f"(per table min {min(marks):.2f} max {max(marks):.2f})", flush=True)
