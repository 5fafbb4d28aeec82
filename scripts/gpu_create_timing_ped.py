"""Create phases (WHAMD_DEBUG_TIMING=1) of a pedigree table: gpu_create_timing_ped.py columns coverage trio|quartet [distrust]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WHAMD_DEBUG_TIMING"] = "1"
from whatshap_amd import _native
from whatshap_amd.blocks import bind_rank_to_device_cpus
from whatshap_amd.synthetic import synthetic_block
bind_rank_to_device_cpus(0, 1, devices=[0])
n, cov, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
p = synthetic_block(n, cov, seed=5, trio=kind == "trio", quartet=kind == "quartet", distrust_genotypes=len(sys.argv) > 4)
for rep in range(3):
    t0 = time.perf_counter(); t = _native.NativeTable(p, solve=False); t1 = time.perf_counter(); t.solve(); t2 = time.perf_counter()
    print("create %.1f ms, solve %.1f ms (device %.1f)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, t.stats()["total_ms"]), flush=True); t.close()
