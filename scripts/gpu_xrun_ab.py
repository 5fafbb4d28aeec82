"""A/B of the register-resident X runs (kernels_slots.h, slot_runx) against the LDS-line runs they replace, on one device: forward time per launch and
the full solution compared.  Usage: gpu_xrun_ab.py [columns] [coverage].  (WHAMD_NO_XRUN is honoured by the debug library only.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
_native.use_debug_library()
from whatshap_amd.synthetic import synthetic_block
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import table_solution

n, cov = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 20
p = synthetic_block(n, cov, seed=3)
out = {}
for name, env in (("x runs", None), ("lds runs", "1")):
    if env is None:
        os.environ.pop("WHAMD_NO_XRUN", None)
    else:
        os.environ["WHAMD_NO_XRUN"] = env
    t = _native.NativeTable(p, solve=False)
    best = None
    for _ in range(4):
        t.solve()
        st = t.stats()
        best = st if best is None or st["forward_ms"] < best["forward_ms"] else best
    out[name] = table_solution(t)
    print(f"{name:9s}: forward {best['forward_ms']:.3f} ms / {best['forward_launches']} launches = {best['forward_ms'] * 1e3 / best['forward_launches']:.3f} us per launch, "
          f"{n / (best['forward_ms'] + best['backtrace_ms']) * 1e3 / 1e6:.3f} M columns/s, cost {t.optimal_score()}", flush=True)
    t.close()
print("identical solutions:", out["x runs"] == out["lds runs"])
