#!/usr/bin/env python3
"""Read selection: whatshap_amd.readselect (whamd_readselection, host C++) next to the REAL whatshap.readselect (oracle/_ref/cy)
on the same synthetic ReadSet (host only; run where oracle/_ref/cy is built).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import build_cython_ref  # noqa: E402
from whatshap_amd import _native  # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    max_cov = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    rng = np.random.default_rng(1)
    n_variants = n_reads // 3
    starts = np.sort(rng.integers(0, n_variants - 30, size=n_reads))
    lengths = rng.integers(2, 25, size=n_reads)
    read_ptr = np.zeros(n_reads + 1, dtype=np.uint64)
    read_ptr[1:] = np.cumsum(lengths)
    pos = np.concatenate([10 * (np.arange(s, s + l) + 1) for s, l in zip(starts, lengths)]).astype(np.int32)
    qual = rng.integers(5, 40, size=pos.size).astype(np.uint32)
    ref = build_cython_ref.import_reference()
    import whatshap.readselect as ref_readselect

    rs = ref.ReadSet()
    for r in range(n_reads):
        read = ref.Read(f"r{r}", 60, 0, 0)
        for i in range(int(read_ptr[r]), int(read_ptr[r + 1])):
            read.add_variant(int(pos[i]), 0, int(qual[i]))
        rs.add(read)
    t0 = time.perf_counter()
    want = set(ref_readselect.readselection(rs, max_cov))
    t1 = time.perf_counter()
    mask = _native.readselection(read_ptr, pos, qual, max_cov)
    t2 = time.perf_counter()
    got = set(np.flatnonzero(mask).tolist())
    print(json.dumps({"reads": n_reads, "variants": int(np.unique(pos).size), "max_cov": max_cov, "selected": len(want), "same_selection": got == want,
                      "reference_s": t1 - t0, "native_s": t2 - t1, "speedup": (t1 - t0) / (t2 - t1), "reads_per_s_native": n_reads / (t2 - t1)}))


if __name__ == "__main__":
    main()
