"""Slice-size sweep of the resident path (resident_l = preferred log2 slice size) for a few coverages."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block
for cov, n in ((15, 50000), (18, 30000), (20, 30000), (12, 50000)):
    p = synthetic_block(n_variants=n, coverage=cov, seed=3)
    for lp in (12, 11, 10, 9, 8, 7):
        t = _native.NativeTable(p, solve=False, path="resident")
        t.set_option("resident_l", str(lp))
        for rep in range(2): t.solve()
        s = t.stats()
        print(f"cov {cov} l_pref {lp}: fwd {s['forward_ms']:.2f} ms bt {s['backtrace_ms']:.2f} ms launches {s['forward_launches']} "
              f"us/col {s['forward_ms']*1e3/s['n_columns']:.3f} cols/s {s['n_columns']/(s['total_ms']/1e3):.0f}", flush=True)
        t.close()
