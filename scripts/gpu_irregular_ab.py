"""Irregular read layout (Poisson starts, geometric lengths): X runs against the LDS-line runs, which runs take which kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
_native.use_debug_library()
from whatshap_amd.synthetic import irregular_block
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import table_solution
p = irregular_block(int(sys.argv[1]) if len(sys.argv) > 1 else 30000, 20, seed=7)
out = {}
for name, env in (("x runs", None), ("lds runs", "1"), ("x runs", None), ("lds runs", "1")):
    os.environ.pop("WHAMD_NO_XRUN", None)
    if env:
        os.environ["WHAMD_NO_XRUN"] = env
    t = _native.NativeTable(p, solve=False)
    best = None
    for _ in range(4):
        t.solve()
        st = t.stats()
        best = st if best is None or st["forward_ms"] < best["forward_ms"] else best
    out[name] = table_solution(t)
    print(f"{name:9s}: forward {best['forward_ms']:.3f} ms / {best['forward_launches']} launches = {best['forward_ms'] * 1e3 / best['forward_launches']:.3f} us per launch, cost {t.optimal_score()}", flush=True)
    t.close()
print("identical solutions:", out["x runs"] == out["lds runs"])
