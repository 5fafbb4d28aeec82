"""Where a step of many tables goes: wall time of whamd_dptable_enqueue_many and of whamd_dptable_wait_many next to the device's forward time.
Usage: gpu_group_step_pieces.py [tables] [coverage] [columns]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WHAMD_DEBUG_TIMING", "1")
from whatshap_amd import _native
from whatshap_amd.blocks import bind_rank_to_device_cpus
bind_rank_to_device_cpus(0, 1, devices=[0])      # (as bench.py does: the whole NUMA node of the GPU, not both sockets)
if os.environ.get("WHAMD_USE_DEBUG_LIB"):
    _native.use_debug_library()                  # (honours the debug switches, e.g. WHAMD_WAIT_ONE_PHASE=1)
from whatshap_amd.synthetic import synthetic_block
nt, cov, n = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 96), (2, 15), (3, 50000)))
problems = [synthetic_block(n, cov, seed=100 + i) for i in range(nt)]
os.environ.pop("WHAMD_DEBUG_TIMING", None)
tables = [_native.NativeTable(p, solve=False, options={"shared_launches": "1"}) for p in problems]
os.environ["WHAMD_DEBUG_TIMING"] = "1"
for rep in range(8):
    t0 = time.perf_counter()
    _native.enqueue_many(tables)
    t1 = time.perf_counter()
    _native.wait_many(tables)
    t2 = time.perf_counter()
    st = [t.stats() for t in tables]
    print(f"rep {rep}: enqueue_many {(t1 - t0) * 1e3:.1f} ms, wait_many {(t2 - t1) * 1e3:.1f} ms; device forward {st[0]['forward_ms']:.1f} ms, backtrace per table {st[0]['backtrace_ms']:.2f} ms (sum {sum(s['backtrace_ms'] for s in st):.1f}), "
          f"host finish per table {st[0]['host_finish_ms']:.2f} ms (sum {sum(s['host_finish_ms'] for s in st):.1f}); device first event to last event of a table (total_ms) min {min(s['total_ms'] for s in st):.1f} max {max(s['total_ms'] for s in st):.1f}", flush=True)
