"""Coverage 21-23 (one table is 512 / 1 024 / 2 048 workgroups per launch): the X kernel with its operands in LDS lines (72 KB per workgroup: two
workgroups per CU) against the streamed variant (16 KB: four per CU), and the eight-cell layout.  Forward time per launch, solutions compared.
Usage: gpu_wide_ab.py [columns]  (debug library: WHAMD_XSTREAM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
_native.use_debug_library()
from whatshap_amd.synthetic import synthetic_block
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import table_solution

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
for cov in (20, 21, 22, 23):
    p = synthetic_block(n, cov, seed=3)
    ref = None
    for name, env, opts in (("lds lines", {}, None), ("streamed", {"WHAMD_XSTREAM": "1"}, None), ("eight cells", {}, {"shared_launches": "1"}),
                            ("eight cells streamed", {"WHAMD_XSTREAM": "1"}, {"shared_launches": "1"})):
        for k in ("WHAMD_XSTREAM",):
            os.environ.pop(k, None)
        os.environ.update(env)
        t = _native.NativeTable(p, solve=False, options=opts)
        best = None
        for _ in range(3):
            t.solve()
            st = t.stats()
            best = st if best is None or st["forward_ms"] < best["forward_ms"] else best
        sol = table_solution(t)
        ref = sol if ref is None else ref
        print(f"coverage {cov} {name:22s}: {best['forward_ms'] * 1e3 / best['forward_launches']:8.2f} us per launch x {best['forward_launches']} launches, "
              f"{n / (best['forward_ms'] + best['backtrace_ms']) * 1e3 / 1e3:8.1f} k columns/s, identical {sol == ref}", flush=True)
        t.close()
