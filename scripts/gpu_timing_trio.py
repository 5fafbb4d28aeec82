"""Timing probe for the trio (config 4) shape: in-kernel cycle split (WHAMD_DEBUG_STAMPS) and slice-size sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
_native.use_debug_library()   # the timing switches and cycle stamps exist in libwhatshap_amd_debug.so only (csrc/debug_build.h)
from whatshap_amd.synthetic import synthetic_block
p = synthetic_block(n_variants=20000, coverage=15, seed=4, trio=True)
print(_native.plan_summary(p))
for path, lp in [("resident", 7), ("resident", 8), ("resident", 6)]:
    t = _native.NativeTable(p, solve=False, path=path)
    if lp is not None: t.set_option("resident_l", str(lp))
    for rep in range(2): t.solve()
    s = t.stats()
    print(path, lp, "fwd %.2fms bt %.2fms launches %d us/col %.3f cols/s %.0f" % (s["forward_ms"], s["backtrace_ms"], s["forward_launches"], s["forward_ms"]*1e3/s["n_columns"], s["n_columns"]/(s["total_ms"]/1e3)), "cost", t.optimal_score(), flush=True)
