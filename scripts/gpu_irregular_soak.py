"""Soak of the irregular layouts on the device against the oracle: single individuals (up to eight reads ending in one column of
a run) and trios (up to four), auto path and the per-column path.  Prints the number of mismatches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from helpers import table_solution, native_solution
from whatshap_amd import _native
from whatshap_amd.synthetic import irregular_block
from test_gpu_parity import _irregular_problem

bad = n = 0
for seed in range(80):
    p = irregular_block(150 + 10 * (seed % 7), 8 + seed % 6, seed=1000 + seed, mean_length=3 + seed % 9)
    want = table_solution(oracle.OracleTable(p))
    for path in ("auto", "column"):
        n += 1
        if native_solution(p, path) != want:
            bad += 1; print("MISMATCH single", seed, path, flush=True)
for seed in range(80):
    p = _irregular_problem(500 + seed, 120, True, 8 + seed % 4)
    want = table_solution(oracle.OracleTable(p))
    for path in ("auto", "resident"):
        n += 1
        if native_solution(p, path) != want:
            bad += 1; print("MISMATCH trio", seed, path, flush=True)
print("irregular soak:", n, "solves,", bad, "mismatches")
sys.exit(1 if bad else 0)
