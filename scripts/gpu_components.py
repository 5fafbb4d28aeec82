"""Chromosome-like single-individual ReadSet made of many connected components, solved as ONE table: lanes (streams
over which the components are spread) 1 vs 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from whatshap_amd import _native
from gpu_multiblock import chromosome

for coverage, n_blocks in ((15, 200), (10, 400), (18, 40), (20, 12)):
    whole = chromosome(n_blocks, coverage, seed=coverage)
    ref = None
    for lanes in (1, 4, 16, 32, 64):
        t = _native.NativeTable(whole, solve=False)
        t.set_option("lanes", str(lanes))
        for rep in range(3):
            t0 = time.perf_counter(); t.solve(); dt = time.perf_counter() - t0
        sol = (t.optimal_score(), t.index_path()[0].tolist(), t.partitioning().tolist())
        if ref is None: ref = sol
        assert sol == ref, "lanes change the result"
        s = t.stats()
        print(f"cov {coverage} {n_blocks} components {t.n_columns} cols lanes {lanes}: solve {dt*1e3:.1f} ms ({t.n_columns/dt:.0f} cols/s), launches {s['forward_launches']}", flush=True)
        t.close()
