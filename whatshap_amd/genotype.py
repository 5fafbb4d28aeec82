"""``GenotypeDPTable`` -- the drop-in for ``whatshap.core.GenotypeDPTable`` (``whatshap/core.pyx:581-602``; its caller is
``whatshap/cli/genotype.py:357-368``), SURVEY.md section 8 row (f3): the forward-backward genotyper over the same columns,
indexing scheme and pedigree partitions as the phasing table (``src/genotypedptable.cpp``).

The whole computation runs on the device behind the C ABI (``whamd_genotype_likelihoods``,
``whatshap_amd/csrc/genotype_device.hip``); there is no CPU fallback.  The reference computes in ``long double``, the
device in f64: likelihoods agree to a relative tolerance (tests: 1e-9), not bit for bit.
"""
from typing import Iterable, Optional

import numpy as np

from . import _native, core


class GenotypeDPTable:
    """``GenotypeDPTable(numeric_sample_ids, readset, recombcost, pedigree, positions=None)`` -- same constructor as the
    reference; like there, the constructor does all the work.  ``pedigree`` must carry genotype likelihoods (the priors)
    for every individual and variant (the reference asserts that, ``src/transitionprobabilitycomputer.cpp:66``)."""

    def __init__(self, numeric_sample_ids, readset, recombcost, pedigree, positions: Optional[Iterable[int]] = None,
                 device: int = 0, window: int = 0, problem: Optional[_native.ProblemArrays] = None):
        self.numeric_sample_ids = numeric_sample_ids
        self.pedigree = pedigree
        if problem is None:
            problem = self._problem(readset, recombcost, pedigree, positions)
        self._problem_arrays = problem
        if problem.positions is not None:
            n_columns = int(problem.positions.size)
        else:
            n_columns = int(np.unique(problem.var_position).size)
        self._individual_ids = [int(x) for x in problem.individual_id]
        self._gl, self._stats = _native.genotype_likelihoods(problem, n_columns, device=device, window=window)

    @staticmethod
    def _problem(readset, recombcost, pedigree, positions):
        if isinstance(pedigree, core.Pedigree):
            return core.problem_from_objects(readset, recombcost, pedigree, False, positions)
        from . import ingest

        compiled = ingest.load()
        if compiled is None:
            raise TypeError("a reference Pedigree needs the compiled ingestion (whatshap_amd.ingest) or whatshap_amd.core.Pedigree")
        return core.problem_from_reference_objects(compiled, readset, recombcost, pedigree, False, positions)

    def get_genotype_likelihoods(self, sample_id, pos: int) -> core.PhredGenotypeLikelihoods:
        numeric = self.numeric_sample_ids[sample_id]
        index = None
        for i in range(len(self._individual_ids) - 1, -1, -1):   # Pedigree::id_to_index: the last individual with that id
            if self._individual_ids[i] == numeric:
                index = i
                break
        if index is None:
            raise RuntimeError(f"Individual with ID {numeric} not present in pedigree.")
        return core.PhredGenotypeLikelihoods([float(x) for x in self._gl[index, pos]])

    def get_stats(self) -> dict:
        return dict(self._stats)
