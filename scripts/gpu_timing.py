"""Timing probe: resident vs column path at the BASELINE shapes (optionally with WHAMD_DEBUG_STAMPS=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
_native.use_debug_library()   # the timing switches and cycle stamps exist in libwhatshap_amd_debug.so only (csrc/debug_build.h)
from whatshap_amd.synthetic import synthetic_block
cases = [dict(n_variants=5000, coverage=15, seed=2), dict(n_variants=4000, coverage=20, seed=3)]
paths = sys.argv[1:] or ["resident", "column"]
for kw in cases:
    p = synthetic_block(**kw)
    for path in paths:
        for lp in ([None] if path != "resident" else [None] + [int(x) for x in os.environ.get("LPREFS", "").split(",") if x]):
            t = _native.NativeTable(p, solve=False, path=path)
            if lp is not None: t.set_option("resident_l", str(lp))
            for rep in range(2):
                t.solve()
            s = t.stats()
            print(path, lp, kw, "fwd %.2fms bt %.2fms total %.2fms launches %d us/col %.3f cols/s %.0f" % (
                s["forward_ms"], s["backtrace_ms"], s["total_ms"], s["forward_launches"], s["forward_ms"]*1e3/s["n_columns"], s["n_columns"]/(s["total_ms"]/1e3)), "cost", t.optimal_score(), flush=True)
