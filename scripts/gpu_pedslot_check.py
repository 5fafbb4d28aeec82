"""First-contact GPU check of the pedigree slot runs (kernels_pedslots.h): device vs oracle on random tie-heavy trios / quartets
and on synthetic blocks (path auto = pedigree slot runs; resident / column for comparison), sequential and chunked backtrace,
then timings of configs[3] (trio, 100 000 columns, coverage 15) and of a quartet table.  Prints one line per case."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from whatshap_amd import _native
_native.use_debug_library()   # the timing switches and cycle stamps exist in libwhatshap_amd_debug.so only (csrc/debug_build.h)
from whatshap_amd.synthetic import random_small_instance, synthetic_block
from oracle import OracleTable, solution_tuple, OracleError


def native_tuple(t):
    a0, a1, q, tv, sid = t.super_reads()
    idx, tv2 = t.index_path()
    return {"cost": t.optimal_score(), "index_path": idx.tolist(), "transmission": tv.tolist(), "path_transmission": tv2.tolist(),
            "partitioning": t.partitioning().tolist(), "allele0": a0.tolist(), "allele1": a1.tolist(), "quality": q.tolist(),
            "sample_ids": sid.tolist(), "positions": t.positions().tolist()}


def first_difference(o, n):
    for k in o:
        if o[k] != n[k]:
            a, b = o[k], n[k]
            if isinstance(a, list) and isinstance(b, list) and len(a) == len(b):
                d = [i for i in range(len(a)) if a[i] != b[i]]
                return f"{k}: {len(d)} of {len(a)} differ, first at {d[:6]}: want {[a[i] for i in d[:6]]} got {[b[i] for i in d[:6]]}"
            return f"{k}: want {str(a)[:120]} got {str(b)[:120]}"
    return "equal"


print("devices", _native.device_count(), flush=True)
bad = 0
quick = "--quick" in sys.argv
for mode in ("trio", "quartet"):
    rng = random.Random(21 if mode == "trio" else 22)
    ok = conf = 0
    for it in range(150 if quick else 300):
        p = random_small_instance(rng, mode=mode)
        try:
            o = solution_tuple(OracleTable(p)); oerr = None
        except OracleError as e:
            oerr = str(e)
        try:
            n = native_tuple(_native.NativeTable(p)); nerr = None
        except _native.SolverError as e:
            nerr = str(e)
        if oerr or nerr:
            if oerr != nerr:
                print("ERR MISMATCH", mode, it, oerr, nerr); bad += 1
            conf += 1
            continue
        if o != n:
            bad += 1
            if bad < 12:
                print("MISMATCH", mode, it, first_difference(o, n), _native.plan_summary(p))
        else:
            ok += 1
    print(mode, "random: ok", ok, "conflicts", conf, "bad so far", bad, flush=True)

cases = [dict(n_variants=200, coverage=7, seed=4, trio=True), dict(n_variants=200, coverage=9, seed=5, trio=True),
         dict(n_variants=300, coverage=10, seed=6, trio=True, mixed_genotypes=True), dict(n_variants=260, coverage=11, seed=7, trio=True, step=1),
         dict(n_variants=300, coverage=12, seed=9, trio=True), dict(n_variants=200, coverage=8, seed=10, quartet=True),
         dict(n_variants=320, coverage=10, seed=11, quartet=True, mixed_genotypes=True), dict(n_variants=150, coverage=6, seed=12, trio=True, distrust_genotypes=True)]
for kw in cases:
    p = synthetic_block(**kw)
    t0 = time.time()
    o = solution_tuple(OracleTable(p))
    t_or = time.time() - t0
    for path in ("auto", "resident" if kw.get("trio") else "column"):
        n = native_tuple(_native.NativeTable(p, path=path))
        d = first_difference(o, n)
        bad += d != "equal"
        print(kw, path, "cost", o["cost"], d, "(oracle %.1fs)" % t_or, flush=True)
# tie-heavy variants (two-valued weights) of the synthetic trio / quartet: the tie rule inside the kernels
for kw in (dict(n_variants=400, coverage=10, seed=31, trio=True), dict(n_variants=300, coverage=9, seed=32, quartet=True)):
    b = synthetic_block(**kw)
    p = _native.ProblemArrays(b.read_ptr, b.var_position, b.var_allele, (1 + (b.var_quality % 2)).astype(np.uint32), b.read_sample_id, b.individual_id,
                              b.triple_ids, b.genotype.reshape(b.n_individuals, b.n_variants), None, (b.recombcost % 3).astype(np.uint32), b.positions, False,
                              n_variants=b.n_variants)
    o = solution_tuple(OracleTable(p))
    n = native_tuple(_native.NativeTable(p))
    d = first_difference(o, n)
    bad += d != "equal"
    print("ties", kw, "cost", o["cost"], d, flush=True)
print("BAD after parity", bad, flush=True)

# chunked vs sequential backtrace and the other paths on a mid-size table (no oracle: the repo's own paths must agree)
for kw in (dict(n_variants=6000, coverage=13, seed=41, trio=True), dict(n_variants=4000, coverage=11, seed=42, quartet=True)):
    p = synthetic_block(**kw)
    ref = None
    for label, path, env in (("auto chunked", "auto", {}), ("auto sequential", "auto", {"WHAMD_BT_SEQUENTIAL": "1"}), ("resident" if kw.get("trio") else "column", "resident" if kw.get("trio") else "column", {})):
        for k, v in env.items():
            os.environ[k] = v
        t = _native.NativeTable(p, path=path)
        for k in env:
            del os.environ[k]
        n = native_tuple(t)
        s = t.stats()
        if ref is None:
            ref = n
        d = first_difference(ref, n)
        bad += d != "equal"
        print(kw, label, "cost", n["cost"], d, "fwd %.2f ms bt %.2f ms launches %d" % (s["forward_ms"], s["backtrace_ms"], s["forward_launches"]), flush=True)

# timings
os.environ["WHAMD_BT_STATS"] = "1"
for kw in (dict(n_variants=100000, coverage=15, seed=4, trio=True), dict(n_variants=50000, coverage=13, seed=5, quartet=True)):
    p = synthetic_block(**kw)
    for path in ("auto", "resident" if kw.get("trio") else "column"):
        if path == "column" and not quick:
            pass
        t0 = time.time()
        t = _native.NativeTable(p, solve=False, path=path)
        create = time.time() - t0
        costs = []
        for rep in range(3):
            t0 = time.time(); t.solve(); dt = time.time() - t0
            s = t.stats()
            costs.append(t.optimal_score())
            print(path, kw, "create %.2fs wall %.3fs fwd %.1f ms bt %.2f ms total %.1f ms launches %d columns/s %.0f" % (
                create, dt, s["forward_ms"], s["backtrace_ms"], s["total_ms"], s["forward_launches"], s["n_columns"] / (s["total_ms"] / 1e3)), "cost", costs[-1], flush=True)
        t.close()
print("BAD", bad)
sys.exit(1 if bad else 0)
