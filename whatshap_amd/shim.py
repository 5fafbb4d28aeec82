"""Glue for running the MI355X solver inside an unmodified WhatsHap.

WhatsHap's ``phase`` command binds ``Pedigree`` and ``PedigreeDPTable`` at import (``whatshap/cli/phase.py:34-42``) and
uses them at ``:604-612`` ("phase" stage) and ``:901-935`` (``create_pedigree``).  The reference ``Pedigree`` is opaque
from Python -- no accessor for the order of individuals or for the trios (``whatshap/core.pxd:22-24``) -- so the shim
swaps in a subclass that forwards every call to the C++ object *and* records it in a ``whatshap_amd.core.Pedigree``;
the table factory then hands the recorded pedigree and the reference's own ``ReadSet`` (iterated through its public
API) to ``whatshap_amd.core.PedigreeDPTable`` and converts the superreads back into reference ``ReadSet`` objects, as
``core.pyx:381-404`` builds them.

    import whatshap.cli.phase as phase, whatshap.core as ref
    from whatshap_amd import shim
    shim.install(phase, ref)          # phase.Pedigree / phase.PedigreeDPTable now run on the GPU
    phase.run_whatshap(...)

Nothing here imports WhatsHap: the reference module objects are passed in.
"""

from __future__ import annotations

from typing import Optional

from . import core as amd


def convert_genotype(genotype) -> amd.Genotype:
    """Reference ``Genotype`` (core.pyx:511-545) -> mirror."""
    return amd.Genotype(list(genotype.as_vector()))


def convert_likelihoods(gl) -> Optional[amd.PhredGenotypeLikelihoods]:
    """Reference ``PhredGenotypeLikelihoods`` (core.pyx:469-504; iterates in genotype-index order) -> mirror."""
    if gl is None:
        return None
    return amd.PhredGenotypeLikelihoods([float(x) for x in gl])


def recording_pedigree_class(reference_pedigree_class):
    """Subclass of the reference's ``Pedigree`` whose instances carry ``.amd``, a ``whatshap_amd.core.Pedigree`` with
    the same individuals and relationships.  Typed Cython arguments (``Pedigree pedigree``) accept the subclass."""

    class RecordingPedigree(reference_pedigree_class):
        # the cdef class allocates in __cinit__(numeric_sample_ids); __init__ receives the same argument
        def __init__(self, numeric_sample_ids):
            self.amd = amd.Pedigree(numeric_sample_ids)

        def add_individual(self, id, genotypes, genotype_likelihoods=None):
            genotypes = list(genotypes)
            gls = list(genotype_likelihoods) if genotype_likelihoods else None
            super().add_individual(id, genotypes, gls)
            self.amd.add_individual(id, [convert_genotype(g) for g in genotypes],
                                    None if gls is None else [convert_likelihoods(g) for g in gls])

        def add_relationship(self, father_id, mother_id, child_id):
            super().add_relationship(father_id, mother_id, child_id)
            self.amd.add_relationship(father_id, mother_id, child_id)

    RecordingPedigree.__name__ = "Pedigree"
    return RecordingPedigree


class _TableAdapter:
    """``PhasingAlgorithm`` interface (``whatshap/types.py:7-15``) over ``whatshap_amd.core.PedigreeDPTable``; superreads
    are returned as the reference's own ReadSet / Read objects when the reference module is known."""

    def __init__(self, table: amd.PedigreeDPTable, reference_core=None):
        self._table = table
        self._ref = reference_core

    def get_optimal_cost(self) -> int:
        return self._table.get_optimal_cost()

    def get_optimal_partitioning(self):
        return self._table.get_optimal_partitioning()

    def get_super_reads(self):
        if self._ref is None:
            return self._table.get_super_reads()
        from . import ingest as _ingest

        compiled = _ingest.load()
        if compiled is not None and hasattr(compiled, "emit_superreads"):
            # compiled emit: the reference builds these ReadSets in C++ and adopts them (core.pyx:388-400,
            # src/pedigreedptable.cpp:344-388); so does whamd_ingest.emit_superreads, from the C-ABI arrays
            positions, a0, a1, q, sid, tv, numbered = self._table.raw_super_reads()
            _counters["compiled_emits"] += 1
            return compiled.emit_superreads(positions, a0, a1, q, sid, numbered), tv.tolist()
        superreads, transmission = self._table.get_super_reads()
        converted = []
        for readset in superreads:
            out = self._ref.ReadSet()
            for read in readset:
                r = self._ref.Read(read.name, -1, -1, read.sample_id)  # core.pyx:390-397: mapq -1, source id -1
                for v in read:
                    r.add_variant(v.position, v.allele, v.quality)
                out.add(r)
            converted.append(out)
        return converted, transmission


# Status codes of the device path's own INPUT limits (include/whatshap_amd.h); the reference has none of them.
# WHAMD_ERR_DEVICE (5: no GPU visible, a HIP runtime fault) is deliberately NOT in the list: a broken installation or a kernel
# fault must surface, not turn into a silent CPU run.
_DEVICE_LIMIT_STATUSES = (4, 6)  # WHAMD_ERR_UNSUPPORTED, WHAMD_ERR_OVERFLOW

_counters = {"device_tables": 0, "cpu_fallbacks": 0, "device_genotype_tables": 0, "cpu_genotype_fallbacks": 0, "compiled_emits": 0,
             "read_selections": 0}
_fallback_reasons = []


def stats() -> dict:
    """How many tables the shim built on the device and how many it handed to the reference class (with the library's
    message for each refusal) since the last ``reset_stats()``."""
    return dict(_counters, fallback_reasons=list(_fallback_reasons))


def reset_stats() -> None:
    for key in _counters:
        _counters[key] = 0
    del _fallback_reasons[:]


def table_factory(reference_core=None, fallback_table_class=None, **solver_options):
    """Callable with the constructor signature of the reference's ``PedigreeDPTable`` (core.pyx:364-379).  The pedigree
    must come from ``recording_pedigree_class`` (or be a ``whatshap_amd.core.Pedigree``).

    ``fallback_table_class`` (OPT-IN: ``install(..., allow_cpu_fallback=True)`` passes the binding it replaces, i.e. the
    reference's own ``PedigreeDPTable``): the device path has input limits the reference does not have -- 25 reads per
    column, 3 trios / 8 individuals per pedigree, 1024 cost terms per column, a pessimistic 32-bit overflow bound.  When
    the library refuses an input for one of those reasons (``WHAMD_ERR_UNSUPPORTED`` / ``_OVERFLOW``) and a fallback class
    was given, the refusal is logged, counted (``shim.stats()``) and the table is built by the fallback class from the
    ORIGINAL ReadSet and pedigree objects.  ``WHAMD_ERR_DEVICE`` -- no GPU visible, a HIP error -- is never a reason to
    fall back: it is re-raised, like the errors of the algorithm itself (Mendelian conflict, unsorted ReadSet), which the
    reference raises too.  Without a fallback class (the default; parity tests and ``bench.py``) every refusal is an error."""

    def make(readset, recombcost, pedigree, distrust_genotypes=False, positions=None):
        # WhatsHap's own objects: compiled ingestion (whatshap_amd/ingest) when it was built -- the C++ ReadSet / Pedigree are
        # walked through thisptr, the pedigree needs no recording subclass; otherwise the objects' public Python API
        from . import ingest as _ingest

        compiled = _ingest.load() if reference_core is not None else None
        problem = None
        if compiled is not None and isinstance(readset, reference_core.ReadSet) and isinstance(pedigree, reference_core.Pedigree):
            problem = amd.problem_from_reference_objects(compiled, readset, recombcost, pedigree, distrust_genotypes, positions)
            recorded = getattr(pedigree, "amd", None)
        else:
            recorded = getattr(pedigree, "amd", pedigree)
            if not isinstance(recorded, amd.Pedigree):
                raise TypeError("the pedigree was not created through whatshap_amd.shim (no recorded individuals / trios) "
                                "and the compiled ingestion (whatshap_amd/ingest/build.py) is not available")
        try:
            table = amd.PedigreeDPTable(readset, recombcost, recorded, distrust_genotypes, positions, problem=problem, **solver_options)
        except RuntimeError as exc:
            status = getattr(exc, "status", None)
            if fallback_table_class is None or status not in _DEVICE_LIMIT_STATUSES:
                raise
            import logging

            logging.getLogger("whatshap_amd").warning(
                "device path refused this table (%s); solving it with the reference PedigreeDPTable", exc)
            _counters["cpu_fallbacks"] += 1
            _fallback_reasons.append(str(exc))
            return fallback_table_class(readset, recombcost, pedigree, distrust_genotypes, positions)
        _counters["device_tables"] += 1
        return _TableAdapter(table, reference_core)

    return make


def heuristic_factory(reference_core=None, fallback_class=None):
    """Callable with the constructor signature of the reference's ``PedMecHeuristic`` (core.pyx:675): ``(readset, recombcost,
    pedigree, row_limit=256, distrust_genotypes=False, positions=None, allow_mutations=True, verbosity=0)``.  Same rules as
    ``table_factory``: compiled ingestion or a recorded pedigree; input-limit refusals go to ``fallback_class`` when one was given."""
    from . import heuristic as _heuristic

    def make(readset, recombcost, pedigree, row_limit=256, distrust_genotypes=False, positions=None, allow_mutations=True, verbosity=0):
        from . import ingest as _ingest

        compiled = _ingest.load() if reference_core is not None else None
        problem = None
        if compiled is not None and isinstance(readset, reference_core.ReadSet) and isinstance(pedigree, reference_core.Pedigree):
            problem = amd.problem_from_reference_objects(compiled, readset, recombcost, pedigree, distrust_genotypes, positions)
            recorded = pedigree
        else:
            recorded = getattr(pedigree, "amd", pedigree)
            if not isinstance(recorded, amd.Pedigree):
                raise TypeError("the pedigree was not created through whatshap_amd.shim and the compiled ingestion is not available")
        try:
            solver = _heuristic.PedMecHeuristic(readset, recombcost, recorded, row_limit, distrust_genotypes, positions, allow_mutations, verbosity, problem=problem)
        except RuntimeError as exc:
            status = getattr(exc, "status", None)
            if fallback_class is None or status not in _DEVICE_LIMIT_STATUSES:
                raise
            import logging

            logging.getLogger("whatshap_amd").warning("device path refused this table (%s); solving it with the reference PedMecHeuristic", exc)
            _counters["cpu_fallbacks"] += 1
            _fallback_reasons.append(str(exc))
            return fallback_class(readset, recombcost, pedigree, row_limit, distrust_genotypes, positions, allow_mutations, verbosity)
        _counters["device_tables"] += 1
        return _HeuristicAdapter(solver, reference_core)

    return make


class _HeuristicAdapter(_TableAdapter):
    """``get_super_reads`` as the reference's own objects (through ``_TableAdapter``), plus ``get_mutations``."""

    def get_mutations(self):
        return self._table.get_mutations()


def readselection_factory():
    """Callable with the signature of ``whatshap.readselect.readselection`` (``readselect.pyx:240``; bound by name in
    ``whatshap/cli/phase.py:43`` and called by ``select_reads``, ``:157-170`` -- which ``whatshap genotype`` imports too,
    ``cli/genotype.py:34``): ``whamd_readselection`` behind it, reference ReadSets flattened by the compiled ingestion."""
    from . import readselect as _readselect

    def readselection(readset, max_cov, preferred_source_ids=None, bridging=True):
        _counters["read_selections"] += 1
        return _readselect.readselection(readset, max_cov, preferred_source_ids, bridging)

    return readselection


class _Previous(tuple):
    """What ``install`` replaced: unpacks as ``(Pedigree, PedigreeDPTable)`` like the tuple earlier versions returned, and
    ``restore()`` puts EVERY rebound name back -- ``PedMecHeuristic`` included."""

    def __new__(cls, module, bindings):
        self = super().__new__(cls, (bindings["Pedigree"], bindings["PedigreeDPTable"]))
        self._module = module
        self.bindings = dict(bindings)
        return self

    def restore(self):
        for name, value in self.bindings.items():
            setattr(self._module, name, value)


def install(phase_module, reference_core=None, allow_cpu_fallback=False, **solver_options):
    """Rebinds ``Pedigree``, ``PedigreeDPTable`` and (where the module has them) ``PedMecHeuristic`` and ``readselection``
    (``whatshap/cli/phase.py:43,163``: the step in front of the table) in ``phase_module`` (normally
    ``whatshap.cli.phase``).  Returns the previous bindings: a tuple ``(Pedigree, PedigreeDPTable)`` with ``.restore()`` (every
    rebound name) and ``.bindings`` (dict).  ``allow_cpu_fallback`` defaults to **False**: an input beyond the device path's INPUT limits
    (more than 25 reads in a column, more than three trios) raises instead of silently running on the class that was replaced; with
    ``True`` such inputs are handed to that class (see ``table_factory``; never a missing GPU or a HIP error)."""
    bindings = {"Pedigree": phase_module.Pedigree, "PedigreeDPTable": phase_module.PedigreeDPTable}
    ref_pedigree = reference_core.Pedigree if reference_core is not None else phase_module.Pedigree
    phase_module.Pedigree = recording_pedigree_class(ref_pedigree)
    phase_module.PedigreeDPTable = table_factory(reference_core, fallback_table_class=bindings["PedigreeDPTable"] if allow_cpu_fallback else None, **solver_options)
    if hasattr(phase_module, "PedMecHeuristic"):   # `--algorithm heuristic` (whatshap/cli/phase.py:589-603)
        bindings["PedMecHeuristic"] = phase_module.PedMecHeuristic
        phase_module.PedMecHeuristic = heuristic_factory(reference_core, bindings["PedMecHeuristic"] if allow_cpu_fallback else None)
    if hasattr(phase_module, "readselection"):     # `select_reads` looks the name up in the module at call time (cli/phase.py:163)
        bindings["readselection"] = phase_module.readselection
        phase_module.readselection = readselection_factory()
    return _Previous(phase_module, bindings)


# ------------------------------------------------------------------------------------------------ whatshap genotype
class _GenotypeAdapter:
    """``get_genotype_likelihoods(sample_id, pos)`` returning the reference's ``PhredGenotypeLikelihoods`` when the reference
    module was handed in (``whatshap/cli/genotype.py:369-383`` passes the result on to ``determine_genotype`` and into the
    variant table)."""

    def __init__(self, table, reference_core):
        self._table = table
        self._ref = reference_core

    def get_genotype_likelihoods(self, sample_id, pos):
        gl = self._table.get_genotype_likelihoods(sample_id, pos)
        if self._ref is None:
            return gl
        return self._ref.PhredGenotypeLikelihoods(gl.as_vector())

    def get_stats(self):
        return self._table.get_stats()


def genotype_table_factory(reference_core=None, fallback_table_class=None, **options):
    """Callable with the constructor signature of the reference's ``GenotypeDPTable`` (core.pyx:581-597):
    ``(numeric_sample_ids, readset, recombcost, pedigree, positions=None)``.  Same rules as ``table_factory``: WhatsHap's
    own objects go through the compiled ingestion (or a recorded pedigree); an input beyond the device path's INPUT limits is
    logged, counted and handed to ``fallback_table_class`` when one was given (opt-in); device errors are re-raised."""
    from . import genotype as _genotype

    def make(numeric_sample_ids, readset, recombcost, pedigree, positions=None):
        from . import ingest as _ingest

        compiled = _ingest.load() if reference_core is not None else None
        problem = None
        if compiled is not None and isinstance(readset, reference_core.ReadSet) and isinstance(pedigree, reference_core.Pedigree):
            problem = amd.problem_from_reference_objects(compiled, readset, recombcost, pedigree, False, positions)
            recorded = pedigree
        else:
            recorded = getattr(pedigree, "amd", pedigree)
            if not isinstance(recorded, amd.Pedigree):
                raise TypeError("the pedigree was not created through whatshap_amd.shim and the compiled ingestion is not available")
        try:
            table = _genotype.GenotypeDPTable(numeric_sample_ids, readset, recombcost, recorded, positions, problem=problem, **options)
        except RuntimeError as exc:
            status = getattr(exc, "status", None)
            if fallback_table_class is None or status not in _DEVICE_LIMIT_STATUSES:
                raise
            import logging

            logging.getLogger("whatshap_amd").warning(
                "device path refused this genotyping table (%s); using the reference GenotypeDPTable", exc)
            _counters["cpu_genotype_fallbacks"] += 1
            _fallback_reasons.append(str(exc))
            return fallback_table_class(numeric_sample_ids, readset, recombcost, pedigree, positions)
        _counters["device_genotype_tables"] += 1
        return _GenotypeAdapter(table, reference_core)

    return make


def install_genotype(genotype_module, reference_core=None, allow_cpu_fallback=False, **options):
    """Rebinds ``GenotypeDPTable`` (and ``Pedigree``, unless the compiled ingestion makes the recording subclass unnecessary)
    in ``genotype_module`` (normally ``whatshap.cli.genotype``, which binds them at ``:20-30`` and uses them at ``:357-368``).
    Returns the previous bindings.  ``allow_cpu_fallback`` as in ``install``."""
    from . import ingest as _ingest

    previous = (genotype_module.Pedigree, genotype_module.GenotypeDPTable)
    if reference_core is None or _ingest.load() is None:
        ref_pedigree = reference_core.Pedigree if reference_core is not None else genotype_module.Pedigree
        genotype_module.Pedigree = recording_pedigree_class(ref_pedigree)
    genotype_module.GenotypeDPTable = genotype_table_factory(reference_core, fallback_table_class=previous[1] if allow_cpu_fallback else None, **options)
    return previous
