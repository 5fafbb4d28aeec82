"""Timeline of blocks.solve_blocks on fresh tables: when each window is created / enqueued / collected, one or two windows on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd.blocks import solve_blocks
from whatshap_amd.synthetic import synthetic_block
n, cols, cov = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
problems = [synthetic_block(cols, cov, seed=100 + i) for i in range(n)]
for rep in range(3):
    for depth, eager in ((1, False), (1, True), (2, True)):
        for window in [int(x) for x in os.environ.get("WHAMD_E2E_WINDOWS", "6,8,12,24").split(",")]:
            trace = []
            t0 = time.perf_counter()
            solved = solve_blocks(problems, max_in_flight=window, windows_on_device=depth, trace=trace, eager_create=eager)
            t1 = time.perf_counter()
            for t in solved:
                t.optimal_score(); t.super_reads(); t.partitioning()
            t2 = time.perf_counter()
            for t in solved:
                t.close()
            line = " ".join(f"{what[:3]}{wi}@{ms:.0f}" for what, wi, ms in trace)
            print(f"rep {rep} depth {depth} eager {int(eager)} window {window:2d}: solve_blocks {1e3 * (t1 - t0):6.1f} ms getters {1e3 * (t2 - t1):5.1f} ms | {line}", flush=True)
