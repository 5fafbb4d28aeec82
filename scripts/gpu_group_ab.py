"""Shared launches with and without the X kernel (slot_groupx): N tables of a coverage in one group, forward time per super-step, checksums against
single-table solves.  Usage: gpu_group_ab.py [tables] [coverage] [columns]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
_native.use_debug_library()
from whatshap_amd.synthetic import synthetic_block

nt, cov, n = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 24), (2, 15), (3, 20000)))
problems = [synthetic_block(n, cov, seed=100 + i) for i in range(nt)]
want = None
for name, env in (("lds runs", "1"), ("x runs", None), ("x by wg", "g"), ("lds runs", "1"), ("x runs", None), ("x by wg", "g")):
    os.environ.pop("WHAMD_NO_XRUN", None)
    os.environ.pop("WHAMD_NO_WARM", None)
    if env == "1":
        os.environ["WHAMD_NO_XRUN"] = env
    os.environ.pop("WHAMD_GROUP_BY_WORKGROUP", None)
    if env == "w":
        os.environ["WHAMD_NO_WARM"] = "1"
    if env == "g":
        os.environ["WHAMD_GROUP_BY_WORKGROUP"] = "1"
    tables = [_native.NativeTable(p, solve=False, options={"shared_launches": "1"}) for p in problems]
    best = None
    walls = []
    for _ in range(8):
        t0 = time.perf_counter()
        _native.enqueue_many(tables)
        _native.wait_many(tables)
        wall = time.perf_counter() - t0
        walls.append(round(wall * 1e3, 1))
        st = tables[0].stats()
        if best is None or st["forward_ms"] < best[0]["forward_ms"]:
            best = (st, wall)
    scores = [t.optimal_score() for t in tables]
    st, wall = best
    print(f"{name:9s}: {nt} tables x {n} columns, coverage {cov}: forward {st['forward_ms']:.3f} ms / {st['forward_launches']} launches = "
          f"{st['forward_ms'] * 1e3 / st['forward_launches']:.3f} us per launch, {nt * n / st['forward_ms'] / 1e3:.2f} M columns/s (forward), walls {walls} ms, backtrace {st['backtrace_ms']:.2f} ms, group_tables {st['group_tables']}", flush=True)
    if want is None:
        want = scores
    elif want != scores:
        print("SCORES DIFFER")
    for t in tables:
        t.close()
singles = [_native.NativeTable(p).optimal_score() for p in problems[:4]]
print("first four against single-table solves:", singles == want[:4])
