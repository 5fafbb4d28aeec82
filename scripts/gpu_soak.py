"""One-off soak: many seeded instances through the device paths vs the C oracle (beyond what the test-suite runs)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from helpers import table_solution, first_difference
from whatshap_amd import _native
from whatshap_amd.synthetic import random_small_instance, synthetic_block

SYMMETRY = os.environ.get("WHAMD_SOAK_SYMMETRY")  # "2": force the complement symmetry onto every run


def device_solution(p, path=None):
    t = _native.NativeTable(p, solve=False, path=path)
    if SYMMETRY:
        t.set_option("symmetry", SYMMETRY)
    t.solve()
    out = table_solution(t)
    t.close()
    return out


t0 = time.time()
bad = 0
rng = random.Random(2026)
n_small = 0
for i in range(1500):
    p = random_small_instance(rng)
    try:
        want = table_solution(oracle.OracleTable(p)); werr = None
    except oracle.OracleError as exc:
        want, werr = None, str(exc)
    for path in ("auto", "column"):
        try:
            got = device_solution(p, path); gerr = None
        except _native.SolverError as exc:
            got, gerr = None, str(exc)
        if got != want or (werr is None) != (gerr is None):
            bad += 1
            print("MISMATCH small", i, path, werr, gerr, first_difference(want or {}, got or {}))
    n_small += 1
print(f"{n_small} random tie-heavy instances x 2 paths: {bad} mismatches, {time.time()-t0:.1f} s", flush=True)
n_mid = 0
for seed in range(int(os.environ.get("WHAMD_SOAK_BLOCKS", "40"))):
    r = np.random.default_rng(seed)
    cov = int(r.integers(6, 15))
    trio = bool(seed % 3 == 0)
    n = int(r.integers(150, 500))
    p = synthetic_block(n, min(cov, 11) if trio else cov, seed=1000 + seed, trio=trio, distrust_genotypes=bool(seed % 5 == 0),
                        step=int(r.integers(1, 4)), error_rate=float(r.uniform(0, 0.2)), drop_rate=float(r.uniform(0, 0.4)))
    want = table_solution(oracle.OracleTable(p))
    got = device_solution(p)
    if got != want:
        bad += 1
        print("MISMATCH mid", seed, cov, trio, first_difference(want, got))
    n_mid += 1
print(f"{n_mid} synthetic blocks (coverage 6-14, steps 1-3, error/drop rates varied, single + trio): total mismatches {bad}, {time.time()-t0:.1f} s")
sys.exit(1 if bad else 0)
