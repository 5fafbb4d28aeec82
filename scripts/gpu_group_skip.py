"""Where the time of a group launch goes: N full-width tables in one launch per super-step, with parts switched off (WHAMD_SLOT_SKIP, results void):
1 no exit stores, 2 no records, 4 one column per run, 8 no ending reads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n_tables = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_cols = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
cov = int(sys.argv[3]) if len(sys.argv) > 3 else 20
skips = [int(x) for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else "0,1,2,3,8,11,15".split(","))]
from whatshap_amd import _native
_native.use_debug_library()   # the timing switches and cycle stamps exist in libwhatshap_amd_debug.so only (csrc/debug_build.h)
from whatshap_amd.synthetic import synthetic_block
problems = [synthetic_block(n_cols, cov, seed=100 + i) for i in range(n_tables)]
for skip in skips:
    if skip:
        os.environ["WHAMD_SLOT_SKIP"] = str(skip)
    else:
        os.environ.pop("WHAMD_SLOT_SKIP", None)
    tables = [_native.NativeTable(p, solve=False) for p in problems]
    for rep in range(2):
        _native.enqueue_many(tables)
        for t in tables:
            t.wait()
    st = tables[0].stats()
    print(f"skip {skip:2d}: {n_tables} tables x {n_cols} columns cov {cov}: forward {st['forward_ms']:.2f} ms, {st['forward_launches']} launches, {st['forward_ms'] * 1e3 / st['forward_launches']:.2f} us per launch, group {st['group_tables']}", flush=True)
    for t in tables:
        t.close()
