"""``PedMecHeuristic`` -- the drop-in for ``whatshap.core.PedMecHeuristic`` (``whatshap/core.pyx:674-734``; C++
``src/pedmecheuristic.cpp``), the beam-search solver ``whatshap phase --algorithm heuristic`` selects
(``whatshap/cli/phase.py:589-603``) when the exact DP cannot afford the coverage.  Same constructor, same four getters;
the solve runs as one persistent kernel on the MI355X (``whatshap_amd/csrc/heuristic_device.hip``) and takes every decision
the reference takes.
"""
from __future__ import annotations

from typing import List, Tuple

from . import _native, core


class PedMecHeuristic:
    """``PedMecHeuristic(readset, recombcost, pedigree, row_limit=256, distrust_genotypes=False, positions=None,
    allow_mutations=True, verbosity=0)`` -- argument order of ``core.pyx:675``.  The ReadSet must be sorted."""

    def __init__(self, readset, recombcost, pedigree, row_limit: int = 256, distrust_genotypes: bool = False, positions=None,
                 allow_mutations: bool = True, verbosity: int = 0, device: int = 0, problem=None):
        if problem is None:
            problem = core.problem_from_objects(readset, recombcost, pedigree, distrust_genotypes, positions)
        self._out = _native.pedmec_heuristic(problem, row_limit=row_limit, allow_mutations=allow_mutations, device=device)
        self.pedigree = pedigree

    def raw_super_reads(self):
        """Same tuple as ``core.PedigreeDPTable.raw_super_reads``: every variant has quality 30 and the reads are not numbered."""
        import numpy as np

        out = self._out
        hap = np.asarray(out["haplotypes"])
        a0 = np.ascontiguousarray(hap[:, 0, :], dtype=np.uint8)
        a1 = np.ascontiguousarray(hap[:, 1, :], dtype=np.uint8)
        return (np.asarray(out["positions"], dtype=np.uint32), a0, a1, np.full(a0.shape, 30, dtype=np.uint32),
                np.asarray(out["sample_ids"]), np.asarray(out["transmission"]), False)

    def get_super_reads(self) -> Tuple[List[core.ReadSet], List[int]]:
        """One ReadSet per sample (two reads ``superread_0`` / ``superread_1``, quality 30, src/pedmecheuristic.cpp:105-121) and
        the transmission value of every column."""
        out = self._out
        results = []
        for s, sample_id in enumerate(out["sample_ids"]):
            rs = core.ReadSet()
            for hap in (0, 1):
                read = core.Read(f"superread_{hap}", -1, -1, int(sample_id))
                for c, position in enumerate(out["positions"]):
                    read.add_variant(int(position), int(out["haplotypes"][s, hap, c]), 30)
                rs.add(read)
            results.append(rs)
        return results, [int(t) for t in out["transmission"]]

    def get_optimal_cost(self) -> int:
        """``int getOptScore()`` (cpp.pxd:265): the reference never assigns its optScore, the value is 0."""
        return int(self._out["score"])

    def get_optimal_partitioning(self) -> List[int]:
        """0 where the bit of getOptBipartition is set, 1 where it is not (core.pyx:711-717)."""
        return [0 if b else 1 for b in self._out["bipartition"]]

    def get_mutations(self):
        """Per sample the (haplotype, column) pairs of alleles that do not follow their parent (core.pyx:719-731)."""
        mut = self._out["mutated"]
        return [[(hap, c) for c in range(mut.shape[2]) for hap in (0, 1) if mut[s, hap, c]] for s in range(mut.shape[0])]

    def get_stats(self) -> dict:
        return self._out["stats"]
