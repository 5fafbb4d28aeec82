"""The drop-in as `whatshap phase` sees it: WhatsHap's OWN ReadSet / Pedigree in (compiled whatshap.core of oracle/_ref/cy), shim table,
reference ReadSets out -- against the native end to end (host arrays in, arrays out) of the same table.  Prints the pieces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import build_cython_ref
ref = build_cython_ref.import_reference()
from whatshap_amd import _native, ingest, shim
from whatshap_amd.synthetic import synthetic_block
from refobjects import problem_to_reference

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 20
trio = len(sys.argv) > 3 and sys.argv[3] == "trio"
p = synthetic_block(n, cov, seed=4 if trio else 3, trio=trio)
t0 = time.perf_counter()
rs, ped = problem_to_reference(p, ref)
print("reference objects built in %.2f s: %d reads, %d variants" % (time.perf_counter() - t0, len(rs), p.var_position.size), flush=True)
recomb, positions = p.recombcost.tolist(), p.positions.tolist()
comp = ingest.load()
make = shim.table_factory(ref)
for rep in range(4):
    t0 = time.perf_counter(); a = comp.flatten_readset(rs); t1 = time.perf_counter(); b = comp.flatten_pedigree(ped); t2 = time.perf_counter()
    comp.u32_array(recomb); comp.u32_array(positions); t3 = time.perf_counter()
    print("  flatten_readset %.1f ms, flatten_pedigree %.1f ms, recomb + positions %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    t0 = time.perf_counter()
    table = make(rs, recomb, ped, False, positions)
    t1 = time.perf_counter()
    sets, tv = table.get_super_reads()
    t2 = time.perf_counter()
    cost = table.get_optimal_cost(); part = table.get_optimal_partitioning()
    t3 = time.perf_counter()
    print("shim: constructor %.1f ms, get_super_reads %.1f ms, cost + partitioning %.1f ms, total %.1f ms -> %.0f columns/s" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, n / (t3 - t0)), flush=True)
    del table, sets
    t0 = time.perf_counter()
    nt = _native.NativeTable(p, solve=False); t1 = time.perf_counter(); nt.solve(); nt.optimal_score(); nt.super_reads(); nt.partitioning()
    t2 = time.perf_counter()
    print("native: create %.1f ms, solve + getters %.1f ms, total %.1f ms -> %.0f columns/s (device %.1f ms)" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, n / (t2 - t0), nt.stats()["total_ms"]), flush=True)
    nt.close()
