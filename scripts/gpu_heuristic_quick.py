"""A few seconds on the device: one PedMecHeuristic table of the bench's shape (rate) and a small one against the same solver source on a host thread."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block
small = synthetic_block(n_variants=300, coverage=26, seed=5)
a, b = _native.pedmec_heuristic(small), _native.pedmec_heuristic(small, host_diagnostic=True)
same = all((a[k] == b[k]).all() if hasattr(a[k], "all") else a[k] == b[k] for k in a if k != "stats")
print("device == host instantiation:", same, flush=True)
big = synthetic_block(n_variants=int(sys.argv[1]) if len(sys.argv) > 1 else 8000, coverage=30, seed=3)
for _ in range(2):
    r = _native.pedmec_heuristic(big)
    print("columns/s", big.n_variants / (r["stats"]["device_ms"] * 1e-3), "device_ms", r["stats"]["device_ms"], flush=True)
