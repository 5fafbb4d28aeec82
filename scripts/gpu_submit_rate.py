"""Is a single table's forward pass bound by the host's launch rate?  Wall time of whamd_dptable_enqueue (the submission of every launch) against the
device's forward time, for a regular and an irregular layout."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
from whatshap_amd.synthetic import irregular_block, synthetic_block
import ctypes as C
L = _native.lib()
for name, p in (("regular cov 20", synthetic_block(30000, 20, seed=3)), ("regular cov 15", synthetic_block(30000, 15, seed=3)), ("irregular cov 20", irregular_block(30000, 20, seed=7))):
    t = _native.NativeTable(p, solve=False)
    t.solve()
    for _ in range(3):
        t0 = time.perf_counter()
        _native._check(L.whamd_dptable_enqueue(t._h))
        t1 = time.perf_counter()
        _native._check(L.whamd_dptable_wait(t._h))
        t2 = time.perf_counter()
        st = t.stats()
        print(f"{name:17s}: enqueue (host) {(t1 - t0) * 1e3:7.2f} ms, wait {(t2 - t1) * 1e3:7.2f} ms, device forward {st['forward_ms']:7.2f} ms, {st['forward_launches']} launches: "
              f"{(t1 - t0) * 1e6 / st['forward_launches']:.2f} us of host per launch, {st['forward_ms'] * 1e3 / st['forward_launches']:.2f} us of device per launch", flush=True)
    t.close()
