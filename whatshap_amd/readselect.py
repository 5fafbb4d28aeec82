"""``readselection`` -- the drop-in for ``whatshap.readselect.readselection`` (``whatshap/readselect.pyx:218-255``; its
caller is ``whatshap/cli/phase.py:157-170`` ``select_reads``), SURVEY.md section 8 row (f2): the step that prunes a
ReadSet to the coverage the DP can afford, directly upstream of ``PedigreeDPTable``.

Host code behind the C ABI (``whamd_readselection``, ``whatshap_amd/csrc/readselect.cpp``): the algorithm is a priority
queue whose scores change after every pick, there is nothing for the GPU in it.  It returns the same *set* of read
indices as the reference, ties included (the C++ replays the iteration orders of the reference's Python sets).
"""
import sys
import warnings
from typing import Iterable, Optional, Set

import numpy as np

from . import _native

# Which read wins a tie depends, in the reference, on the layout of CPython `set`s and of a libstdc++ unordered_set; the C++
# side replays CPython 3.8 - 3.12's setobject.c (unchanged across those versions) and was validated against the built
# reference under CPython 3.10 (tests/test_readselect.py).  Outside that range the selection is still a valid one -- same
# coverage bound, same scores -- but tie order is not guaranteed to match an installed WhatsHap.
_VALIDATED_PYTHONS = ((3, 8), (3, 12))
_warned = False


def _check_interpreter():
    global _warned
    if _warned or _VALIDATED_PYTHONS[0] <= sys.version_info[:2] <= _VALIDATED_PYTHONS[1]:
        return
    _warned = True
    warnings.warn(f"whatshap_amd.readselect: tie order was validated for CPython {_VALIDATED_PYTHONS[0]} .. {_VALIDATED_PYTHONS[1]}, "
                  f"this is {sys.version_info[:2]}: the selected set may differ from whatshap.readselect's on ties", RuntimeWarning, stacklevel=3)


def _flat(readset):
    """(read_ptr, positions, qualities) of our ReadSet mirror or of a reference ReadSet."""
    from . import core

    if isinstance(readset, core.ReadSet):
        read_ptr, pos, _alle, qual, _samples = core._flatten_readset(readset)
        return read_ptr, pos, qual
    from . import ingest

    compiled = ingest.load()
    if compiled is not None:
        try:
            read_ptr, pos, _alle, qual, _samples = compiled.flatten_readset(readset)
            return read_ptr, pos, qual
        except TypeError:
            pass
    read_ptr = [0]
    pos, qual = [], []
    for read in readset:
        for v in read:
            pos.append(v.position)
            qual.append(v.quality)
        read_ptr.append(len(pos))
    return np.asarray(read_ptr, dtype=np.uint64), np.asarray(pos, dtype=np.int32), np.asarray(qual, dtype=np.uint32)


def _source_ids(readset) -> np.ndarray:
    """``read.source_id`` of every read; reference ReadSets through the compiled ingestion (one call instead of one per read)."""
    from . import core, ingest

    if not isinstance(readset, core.ReadSet):
        compiled = ingest.load()
        if compiled is not None and hasattr(compiled, "read_source_ids"):
            try:
                return compiled.read_source_ids(readset)
            except TypeError:
                pass
    return np.asarray([read.source_id for read in readset], dtype=np.int32)


def readselection(readset, max_cov: int, preferred_source_ids: Optional[Iterable[int]] = None, bridging: bool = True) -> Set[int]:
    """Indices of the reads to keep so that no variant is covered more than ``max_cov`` times.

    Same signature, result and error as ``whatshap.readselect.readselection``: reads that cover fewer than two variants
    raise ``ValueError`` (``readselect.pyx:236-239``).
    """
    _check_interpreter()
    read_ptr, pos, qual = _flat(readset)
    sources = None
    if preferred_source_ids is not None:
        preferred_source_ids = set(int(s) for s in preferred_source_ids)
        sources = _source_ids(readset)
    try:
        mask = _native.readselection(read_ptr, pos, qual, max_cov, sources, preferred_source_ids, bridging)
    except _native.SolverError as e:
        if e.status == _native.WHAMD_ERR_INVALID:
            raise ValueError(str(e)) from None
        raise
    return set(np.flatnonzero(mask).tolist())
