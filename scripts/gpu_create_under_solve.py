"""Does a solve on the device slow the creates of the next window down?  96 creates (16 workers x 2 threads) alone, then the same while another thread keeps
a resident group of tables solving; and the resident step alone / under the creates.  Usage: gpu_create_under_solve.py [tables columns coverage]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from whatshap_amd import _native
from whatshap_amd.blocks import bind_rank_to_device_cpus
from whatshap_amd.synthetic import synthetic_block
if os.environ.get("WHAMD_USE_DEBUG_LIB"):
    _native.use_debug_library()   # (honours the debug switches, e.g. WHAMD_UPLOAD_ON_TABLE_STREAM=1)
k, n, cov = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (96, 50000, 15)
bind_rank_to_device_cpus(0, 1, devices=[0])
problems = [synthetic_block(n, cov, seed=100 + i) for i in range(k)]
opts = {"shared_launches": "1", "host_threads": "2"}
resident = [_native.NativeTable(p, solve=False, options=opts) for p in problems[:k // 2]]

def creates():
    with ThreadPoolExecutor(max_workers=16) as pool:
        t0 = time.perf_counter()
        made = list(pool.map(lambda pr: _native.NativeTable(pr, solve=False, options=opts), problems))
        wall = time.perf_counter() - t0
    for t in made:
        t.close()
    return wall

def step():
    t0 = time.perf_counter()
    _native.enqueue_many(resident)
    _native.wait_many(resident)
    return time.perf_counter() - t0

creates(); step()
for rep in range(3):
    alone = creates()
    s_alone = min(step() for _ in range(3))
    stop = threading.Event()
    steps = []
    def loop():
        while not stop.is_set():
            steps.append(step())
    th = threading.Thread(target=loop)
    th.start()
    time.sleep(0.05)
    under = creates()
    stop.set(); th.join()
    print(f"rep {rep}: {k} creates alone {alone * 1e3:.1f} ms ({k / alone:.0f}/s), under a running solve {under * 1e3:.1f} ms ({k / under:.0f}/s); "
          f"a step of {len(resident)} resident tables alone {s_alone * 1e3:.1f} ms, under the creates {1e3 * sum(steps) / max(len(steps), 1):.1f} ms ({len(steps)} steps)", flush=True)
