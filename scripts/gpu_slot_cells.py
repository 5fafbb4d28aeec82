"""Cells per thread of the single-individual slot runs (option slot_r = 1 / 2 / 3 -> 2 / 4 / 8 cells): parity against the oracle on
tie-heavy instances, agreement of the variants on a mid-size table, timings on configs[2]-shaped tables."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from whatshap_amd import _native
from whatshap_amd.synthetic import random_small_instance, synthetic_block
from oracle import OracleTable


def solve(p, **options):
    t = _native.NativeTable(p, solve=False)
    for k, v in options.items():
        t.set_option(k, str(v))
    t.solve()
    return t


bad = 0
rng = random.Random(5)
for it in range(120):
    p = random_small_instance(rng, mode="single", allow_conflict=False)
    o = OracleTable(p)
    t = solve(p, slot_r=1)
    idx, _ = t.index_path()
    want, _ = o.index_path()
    if t.optimal_score() != o.optimal_score() or not (idx == want).all():
        bad += 1
        if bad < 5:
            print("MISMATCH", it, t.optimal_score(), o.optimal_score())
    t.close()
print("random tie-heavy: bad", bad, flush=True)
for kw in (dict(n_variants=3000, coverage=12, seed=3), dict(n_variants=20000, coverage=20, seed=3), dict(n_variants=8000, coverage=17, seed=4, step=2)):
    p = synthetic_block(**kw)
    ref = None
    for options in (dict(slot_r=2), dict(slot_r=1), dict(slot_r=1, slot_l=9), dict(slot_r=1, symmetry=0)):
        t = solve(p, **options)
        idx, _ = t.index_path()
        got = (t.optimal_score(), idx.tolist(), t.partitioning().tolist())
        if ref is None:
            ref = got
        same = got == ref
        bad += not same
        s = t.stats()
        print(kw, options, "same" if same else "DIFFERENT", "cost", got[0], "fwd %.2f ms launches %d" % (s["forward_ms"], s["forward_launches"]), flush=True)
        t.close()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
p = synthetic_block(n, 20, seed=3)
for options in (dict(slot_r=2), dict(slot_r=1), dict(slot_r=1, slot_l=9)):
    t = _native.NativeTable(p, solve=False)
    for k, v in options.items():
        t.set_option(k, str(v))
    for rep in range(3):
        t.solve()
        s = t.stats()
        print(options, "fwd %.2f ms bt %.2f ms total %.2f ms launches %d -> %.0f columns/s, %.2f us/launch" % (
            s["forward_ms"], s["backtrace_ms"], s["total_ms"], s["forward_launches"], n / (s["total_ms"] / 1e3), s["forward_ms"] * 1e3 / s["forward_launches"]), "cost", t.optimal_score(), flush=True)
    t.close()
print("BAD", bad)
sys.exit(1 if bad else 0)
