"""A/B of the pedigree X runs (kernels_pedslots.h, pedslot_runx) against the LDS-line runs: trio, quartet, trio with untrusted genotypes.
Usage: gpu_pedx_ab.py [columns]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
_native.use_debug_library()
from whatshap_amd.synthetic import synthetic_block
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import table_solution
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for label, kw in (("trio cov 15", dict(coverage=15, seed=4, trio=True)), ("quartet cov 13", dict(coverage=13, seed=5, quartet=True)),
                  ("trio cov 15, genotypes not trusted", dict(coverage=15, seed=4, trio=True, distrust_genotypes=True)), ("trio cov 11", dict(coverage=11, seed=6, trio=True))):
    p = synthetic_block(n, **kw)
    out = {}
    os.environ["WHAMD_PED_XRUN"] = "1"
    for name, env in (("x runs", None), ("lds runs", "1")):
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
        print(f"{label:36s} {name:9s}: forward {best['forward_ms']:.3f} ms / {best['forward_launches']} launches = {best['forward_ms'] * 1e3 / best['forward_launches']:.3f} us per launch, "
              f"{n / (best['forward_ms'] + best['backtrace_ms']) * 1e3 / 1e6:.3f} M columns/s, cost {t.optimal_score()}", flush=True)
        t.close()
    print(f"{label:36s} identical solutions:", out["x runs"] == out["lds runs"], flush=True)
