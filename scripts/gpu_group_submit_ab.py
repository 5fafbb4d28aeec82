"""Host side of a group submission (DeviceTable::enqueue_group): N coverage-15 tables, enqueue_many + wait_many timed, with the compact per-step records of round 5 and
with the old memory traffic added back (WHAMD_GROUP_TOUCH_ENTRIES=1, debug library).  WHAMD_DEBUG_TIMING prints the submission time of the launch sequence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WHAMD_DEBUG_TIMING"] = "1"
from whatshap_amd import _native
_native.use_debug_library()
from whatshap_amd.synthetic import synthetic_block
n_tables, cols = int(sys.argv[1]) if len(sys.argv) > 1 else 96, int(sys.argv[2]) if len(sys.argv) > 2 else 50000
ps = [synthetic_block(cols, 15, seed=100 + i) for i in range(n_tables)]
tables = [_native.NativeTable(p, solve=False, options={"shared_launches": "1"}) for p in ps]
for rep in range(3):
    for name, env in (("briefs", None), ("briefs + the old reads", "1")):
        os.environ.pop("WHAMD_GROUP_TOUCH_ENTRIES", None)
        if env:
            os.environ["WHAMD_GROUP_TOUCH_ENTRIES"] = env
        t0 = time.perf_counter()
        _native.enqueue_many(tables)
        t1 = time.perf_counter()
        _native.wait_many(tables)
        t2 = time.perf_counter()
        st = tables[0].stats()
        print(f"rep {rep} {name:24s}: enqueue {1e3 * (t1 - t0):6.2f} ms, wait {1e3 * (t2 - t1):6.2f} ms, total {1e3 * (t2 - t0):6.2f} ms = {n_tables * cols / (t2 - t0) / 1e6:.1f} M columns/s; forward {st['forward_ms']:.2f} ms / {st['forward_launches']} launches", flush=True)
