#!/usr/bin/env python3
"""Chromosome-like single-individual ReadSet made of many disconnected blocks: one table for everything vs. the
host-side work queue (split at read-free boundaries, blocks in flight on their own streams)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
from whatshap_amd.blocks import split_independent_blocks, solve_blocks
from whatshap_amd.synthetic import synthetic_block


def chromosome(n_blocks, coverage, seed, max_len=1500):
    rng = np.random.default_rng(seed)
    parts = [synthetic_block(int(rng.integers(min(50, max_len // 2), max_len)), coverage, seed=seed * 1000 + b) for b in range(n_blocks)]
    ptr, pos, al, q, positions, geno, recomb = [np.zeros(1, np.uint64)], [], [], [], [], [], []
    offset, base = 0, 0
    for p in parts:
        pos.append(p.var_position + offset); al.append(p.var_allele); q.append(p.var_quality)
        ptr.append(p.read_ptr[1:] + np.uint64(base)); base += int(p.read_ptr[-1])
        positions.append(p.positions + offset); geno.append(p.genotype.reshape(1, -1)); recomb.append(p.recombcost)
        offset = int(positions[-1][-1]) + 1000
    ptr = np.concatenate(ptr)
    return _native.ProblemArrays(ptr, np.concatenate(pos), np.concatenate(al), np.concatenate(q),
                                 np.zeros(ptr.size - 1, np.int32), [0], [], np.concatenate(geno, axis=1), None,
                                 np.concatenate(recomb), np.concatenate(positions), False)


if __name__ == "__main__":
  for coverage, n_blocks in ((15, 200), (10, 400), (18, 40)):
      whole = chromosome(n_blocks, coverage, seed=coverage)
      t0 = time.perf_counter(); table = _native.NativeTable(whole, solve=False); t1 = time.perf_counter()
      table.solve(); t2 = time.perf_counter()
      cost = table.optimal_score(); ncols = table.n_columns
      t3 = time.perf_counter(); blocks = split_independent_blocks(whole); t4 = time.perf_counter()
      for window in (1, 4, 8, 16):
          t5 = time.perf_counter(); tables = solve_blocks([b[0] for b in blocks], max_in_flight=window); t6 = time.perf_counter()
          assert sum(t.optimal_score() for t in tables) == cost
          print(f"cov {coverage} {n_blocks} blocks {ncols} cols: one table create {t1-t0:.3f}s solve {t2-t1:.3f}s | "
                f"split {t4-t3:.3f}s, queue window {window}: create+solve {t6-t5:.3f}s", flush=True)
          del tables
