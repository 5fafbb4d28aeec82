"""A/B of the pedigree runs' min-plus step on packed keys (kernels_pedslots.h, SlotRun::yflags bit 4) against the staged comparison (WHAMD_NO_PED_KEYS=1, debug library):
trio, quartet, both with untrusted genotypes; eight trio tables sharing their launches.  Usage: gpu_pedkeys_ab.py [columns]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
_native.use_debug_library()
from whatshap_amd.synthetic import synthetic_block
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import table_solution
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for label, kw in (("trio cov 15", dict(coverage=15, seed=4, trio=True)), ("quartet cov 13", dict(coverage=13, seed=5, quartet=True)),
                  ("trio cov 15, genotypes not trusted", dict(coverage=15, seed=4, trio=True, distrust_genotypes=True)),
                  ("quartet cov 13, genotypes not trusted", dict(coverage=13, seed=5, quartet=True, distrust_genotypes=True))):
    p = synthetic_block(n, **kw)
    out = {}
    for name, env in (("keys", None), ("staged", "1")):
        os.environ.pop("WHAMD_NO_PED_KEYS", None)
        if env:
            os.environ["WHAMD_NO_PED_KEYS"] = env
        t = _native.NativeTable(p, solve=False)
        best = None
        for _ in range(4):
            t.solve()
            st = t.stats()
            best = st if best is None or st["forward_ms"] < best["forward_ms"] else best
        out[name] = table_solution(t)
        print(f"{label:40s} {name:7s}: forward {best['forward_ms']:.3f} ms / {best['forward_launches']} launches = {best['forward_ms'] * 1e3 / best['forward_launches']:.3f} us per launch, "
              f"{n / (best['forward_ms'] + best['backtrace_ms']) * 1e3 / 1e6:.3f} M columns/s, cost {t.optimal_score()}", flush=True)
        t.close()
    print(f"{label:40s} identical solutions:", out["keys"] == out["staged"], flush=True)
# eight trio tables as one sequence of launches
ps = [synthetic_block(n, coverage=15, seed=40 + i, trio=True) for i in range(8)]
for name, env in (("keys", None), ("staged", "1")):
    os.environ.pop("WHAMD_NO_PED_KEYS", None)
    if env:
        os.environ["WHAMD_NO_PED_KEYS"] = env
    tables = [_native.NativeTable(p, solve=False) for p in ps]
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        _native.enqueue_many(tables)
        _native.wait_many(tables)
        best = min(best, time.perf_counter() - t0)
    st = tables[0].stats()
    print(f"8 trio tables, one group                 {name:7s}: {best * 1e3:.2f} ms wall = {8 * n / best / 1e6:.3f} M columns/s; forward {st['forward_ms']:.2f} ms / {st['forward_launches']} launches = {st['forward_ms'] * 1e3 / st['forward_launches']:.2f} us per launch; costs {sum(t.optimal_score() for t in tables)}", flush=True)
    for t in tables:
        t.close()
