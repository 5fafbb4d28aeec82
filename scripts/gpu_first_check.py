"""First-contact GPU check: HIP path vs oracle on random + synthetic instances, then raw timings."""
import random, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
from whatshap_amd.synthetic import random_small_instance, synthetic_block
from oracle import OracleTable, solution_tuple, OracleError

def native_tuple(t):
    a0, a1, q, tv, sid = t.super_reads()
    idx, tv2 = t.index_path()
    return {"cost": t.optimal_score(), "index_path": idx.tolist(), "transmission": tv.tolist(), "path_transmission": tv2.tolist(),
            "partitioning": t.partitioning().tolist(), "allele0": a0.tolist(), "allele1": a1.tolist(), "quality": q.tolist(),
            "sample_ids": sid.tolist(), "positions": t.positions().tolist()}

print("devices", _native.device_count())
rng = random.Random(11)
bad = 0
for path in ("resident", "column", "column_keys"):
    ok = conf = 0
    for it in range(400):
        p = random_small_instance(rng)
        try:
            o = solution_tuple(OracleTable(p)); oerr = None
        except OracleError as e:
            oerr = str(e)
        try:
            n = native_tuple(_native.NativeTable(p, path=path)); nerr = None
        except _native.SolverError as e:
            nerr = str(e)
        if oerr or nerr:
            if oerr != nerr: print("ERR MISMATCH", it, oerr, nerr); bad += 1
            conf += 1; continue
        if o != n:
            bad += 1
            for k in o:
                if o[k] != n[k]: print("MISMATCH", path, it, k, o[k], n[k]); break
        else: ok += 1
    print(path, "random ok", ok, "conflicts", conf, "bad", bad)

for kw in [dict(n_variants=300, coverage=8, seed=2), dict(n_variants=200, coverage=9, seed=4, trio=True),
           dict(n_variants=150, coverage=6, seed=5, trio=True, distrust_genotypes=True),
           dict(n_variants=400, coverage=10, seed=7, distrust_genotypes=True), dict(n_variants=600, coverage=12, seed=3),
           dict(n_variants=300, coverage=12, seed=9, trio=True), dict(n_variants=120, coverage=14, seed=13, step=1)]:
    p = synthetic_block(**kw)
    o = solution_tuple(OracleTable(p))
    for path in ("resident", "column", "column_keys"):
        n = native_tuple(_native.NativeTable(p, path=path))
        eq = o == n
        if not eq:
            bad += 1
            for k in o:
                if o[k] != n[k]: print("MISMATCH", k, str(o[k])[:200], str(n[k])[:200]); break
        print(kw, path, "cost", o["cost"], "equal", eq)

for kw in [dict(n_variants=5000, coverage=15, seed=2), dict(n_variants=4000, coverage=20, seed=3), dict(n_variants=2000, coverage=15, seed=4, trio=True)]:
    p = synthetic_block(**kw)
    for path in ("resident", "column"):
        t = _native.NativeTable(p, solve=False, path=path)
        for rep in range(2):
            t0 = time.time(); t.solve(); dt = time.time() - t0
            s = t.stats()
            print(path, kw, "wall %.3fs fwd %.1fms bt %.1fms total %.1fms launches %d cols/s %.0f cells/s %.2fG algGB/s %.1f" % (
                dt, s["forward_ms"], s["backtrace_ms"], s["total_ms"], s["forward_launches"], s["n_columns"]/(s["total_ms"]/1e3),
                s["n_cells"]/(s["total_ms"]/1e3)/1e9, s["algorithmic_bytes"]/(s["total_ms"]/1e3)/1e9), "cost", t.optimal_score())
print("BAD", bad)
sys.exit(1 if bad else 0)
