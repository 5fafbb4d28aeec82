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
for i in range(int(os.environ.get("WHAMD_SOAK_SMALL", "1500"))):
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
print(f"{n_mid} synthetic blocks (coverage 6-14, steps 1-3, error/drop rates varied, single + trio): total mismatches {bad}, {time.time()-t0:.1f} s", flush=True)
# high coverage, single individual, DEFAULT symmetry unless WHAMD_SOAK_SYMMETRY is set: full-chip (and halved) runs chained
# through the exchange layouts -- where the benchmark runs.  Prefixes: the ramp (2 * step... columns) + 30-70 full-width columns.
n_high = 0
for seed in range(int(os.environ.get("WHAMD_SOAK_HIGH", "15"))):
    r = np.random.default_rng(7000 + seed)
    cov = int(r.integers(16, 21))
    step = int(r.integers(1, 4))
    ncols = step * cov + int(r.integers(30, 70)) if cov < 20 else step * cov + int(r.integers(30, 50))
    p = synthetic_block(100000, cov, seed=2000 + seed, step=step, distrust_genotypes=bool(seed % 6 == 5), error_rate=float(r.uniform(0, 0.1)),
                        drop_rate=float(r.uniform(0, 0.3)), n_columns_limit=ncols)
    want = table_solution(oracle.OracleTable(p))
    for path in ("auto", "resident"):
        got = device_solution(p, path)
        if got != want:
            bad += 1
            print("MISMATCH high", seed, cov, step, path, first_difference(want, got))
    n_high += 1
    print(f"  high-coverage block {seed}: coverage {cov}, step {step}, {ncols} columns ok={bad == 0} ({time.time()-t0:.0f} s)", flush=True)
print(f"{n_high} high-coverage prefixes (coverage 16-20, steps 1-3, symmetry {SYMMETRY or 'default'}): total mismatches {bad}, {time.time()-t0:.1f} s")
sys.exit(1 if bad else 0)
