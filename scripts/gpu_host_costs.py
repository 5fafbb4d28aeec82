"""Where the wall time of a whole PedigreeDPTable construction goes (host flattening / planning / upload vs solve)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block
for kw in [dict(n_variants=200000, coverage=20, seed=3), dict(n_variants=100000, coverage=15, seed=4, trio=True), dict(n_variants=50000, coverage=15, seed=2)]:
    t0 = time.perf_counter(); p = synthetic_block(**kw); t1 = time.perf_counter()
    t = _native.NativeTable(p, solve=False); t2 = time.perf_counter()
    t.solve(); t3 = time.perf_counter()
    t.solve(); t4 = time.perf_counter()
    s = t.stats()
    print(kw, "generate %.2fs create %.2fs (host_prepare_ms %.0f) first solve %.3fs second solve %.3fs (device total %.1f ms, host finish %.1f ms)" % (
        t1 - t0, t2 - t1, s["host_prepare_ms"], t3 - t2, t4 - t3, s["total_ms"], s["host_finish_ms"]), flush=True)
