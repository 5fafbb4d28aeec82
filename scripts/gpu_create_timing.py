"""Host create path of one configs[2] table (WHAMD_DEBUG_TIMING=1): flatten / plan / descriptors + copies / arena, three times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WHAMD_DEBUG_TIMING"] = "1"
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block

p = synthetic_block(int(sys.argv[1]) if len(sys.argv) > 1 else 200000, int(sys.argv[2]) if len(sys.argv) > 2 else 20, seed=3)
print("host threads", os.cpu_count(), "WHAMD_THREADS", os.environ.get("WHAMD_THREADS"), flush=True)
for rep in range(4):
    t0 = time.perf_counter()
    t = _native.NativeTable(p, solve=False)
    t1 = time.perf_counter()
    t.solve()
    t.optimal_score(), t.super_reads(), t.partitioning()
    t2 = time.perf_counter()
    print("create %.1f ms, solve + getters %.1f ms (device total %.1f ms)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, t.stats()["total_ms"]), flush=True)
    t.close()
