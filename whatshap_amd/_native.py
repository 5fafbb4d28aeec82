"""ctypes binding of the C ABI in include/whatshap_amd.h.

This is the *only* place the shared library is loaded.  There is no CPU fallback: if
``libwhatshap_amd.so`` has not been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C whatshap_amd/csrc``) importing this module raises, and creating a table on a machine
without a HIP device raises ``RuntimeError`` from the library's own message.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwhatshap_amd.so")
DEBUG_LIB_PATH = os.path.join(_HERE, "libwhatshap_amd_debug.so")   # test infrastructure (debug_lib)
ABI_VERSION = 2   # WHAMD_ABI_VERSION of include/whatshap_amd.h this file mirrors (struct layouts below)

WHAMD_OK = 0
WHAMD_ERR_INVALID = 1
WHAMD_ERR_MENDELIAN_CONFLICT = 2
WHAMD_ERR_UNSORTED = 3
WHAMD_ERR_UNSUPPORTED = 4
WHAMD_ERR_DEVICE = 5
WHAMD_ERR_OVERFLOW = 6
WHAMD_ERR_HOST = 7
GT_OTHER = 255


class ReadSetView(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint32),
        ("read_ptr", C.POINTER(C.c_uint64)),
        ("var_position", C.POINTER(C.c_int32)),
        ("var_allele", C.POINTER(C.c_uint8)),
        ("var_quality", C.POINTER(C.c_uint32)),
        ("read_sample_id", C.POINTER(C.c_int32)),
    ]


class PedigreeView(C.Structure):
    _fields_ = [
        ("n_individuals", C.c_uint32),
        ("individual_id", C.POINTER(C.c_uint32)),
        ("n_triples", C.c_uint32),
        ("triple_ids", C.POINTER(C.c_uint32)),
        ("n_variants", C.c_uint32),
        ("genotype", C.POINTER(C.c_uint8)),
        ("genotype_likelihoods", C.POINTER(C.c_double)),
        ("gl_present", C.POINTER(C.c_uint8)),
    ]


class PlanSummary(C.Structure):
    _fields_ = [(name, C.c_uint64) for name in (
        "n_columns", "n_steps", "n_runs", "n_resident_columns", "n_folded_columns", "n_vectorised_columns",
        "max_run_columns", "max_workgroups", "max_lds_bytes", "backtrace_bytes", "n_components", "n_halved_runs")] + [
        ("max_coverage", C.c_uint32), ("invariants_ok", C.c_uint32), ("n_yform_runs", C.c_uint64), ("n_fact_runs", C.c_uint64)]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class SolveStats(C.Structure):
    _fields_ = [
        ("n_columns", C.c_uint64),
        ("n_cells", C.c_uint64),
        ("n_costs", C.c_uint64),
        ("algorithmic_bytes", C.c_uint64),
        ("forward_launches", C.c_uint64),
        ("forward_ms", C.c_double),
        ("backtrace_ms", C.c_double),
        ("total_ms", C.c_double),
        ("host_prepare_ms", C.c_double),
        ("host_finish_ms", C.c_double),
        ("max_coverage", C.c_uint32),
        ("transmissions", C.c_uint32),
        ("bt_chunks", C.c_uint32),
        ("bt_missed", C.c_uint32),
        ("bt_rewalked", C.c_uint32),
        ("group_tables", C.c_uint32),
        ("host_flatten_ms", C.c_double),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "pad"}


class GenotypeStats(C.Structure):
    _fields_ = [
        ("n_columns", C.c_uint64),
        ("n_cells", C.c_uint64),
        ("launches", C.c_uint64),
        ("backward_ms", C.c_double),
        ("forward_ms", C.c_double),
        ("total_ms", C.c_double),
        ("host_prepare_ms", C.c_double),
        ("window", C.c_uint32),
        ("max_coverage", C.c_uint32),
        ("transmissions", C.c_uint32),
        ("slot_runs", C.c_uint32),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "pad"}


class HeuristicStats(C.Structure):
    _fields_ = [
        ("n_columns", C.c_uint64),
        ("n_reads", C.c_uint64),
        ("max_solutions", C.c_uint64),
        ("total_solutions", C.c_uint64),
        ("device_ms", C.c_double),
        ("host_prepare_ms", C.c_double),
        ("host_finish_ms", C.c_double),
        ("n_samples", C.c_uint32),
        ("row_limit", C.c_uint32),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class HeuristicJob(C.Structure):
    _fields_ = [("readset", C.POINTER(ReadSetView)), ("recombcost", C.POINTER(C.c_uint32)), ("n_recombcost", C.c_size_t),
                ("pedigree", C.POINTER(PedigreeView)), ("distrust_genotypes", C.c_int), ("positions", C.POINTER(C.c_uint32)),
                ("n_positions", C.c_size_t), ("row_limit", C.c_uint32), ("allow_mutations", C.c_int)]


def _ptr(arr: Optional[np.ndarray], ctype):
    if arr is None:
        return C.cast(None, C.POINTER(ctype))
    return arr.ctypes.data_as(C.POINTER(ctype))


class ProblemArrays:
    """Owns the numpy arrays behind a (readset view, pedigree view, recombcost, positions) tuple.

    The same object feeds the product library, the oracle restatement and the compiled reference
    driver, so all three see byte-identical inputs.
    """

    def __init__(
        self,
        read_ptr,
        var_position,
        var_allele,
        var_quality,
        read_sample_id,
        individual_id,
        triple_ids,
        genotype,
        genotype_likelihoods,
        recombcost,
        positions,
        distrust_genotypes: bool,
        n_variants: Optional[int] = None,
    ):
        self.read_ptr = np.ascontiguousarray(read_ptr, dtype=np.uint64)
        if self.read_ptr.size == 0:
            self.read_ptr = np.zeros(1, dtype=np.uint64)
        self.var_position = np.ascontiguousarray(var_position, dtype=np.int32)
        self.var_allele = np.ascontiguousarray(var_allele, dtype=np.uint8)
        self.var_quality = np.ascontiguousarray(var_quality, dtype=np.uint32)
        self.read_sample_id = np.ascontiguousarray(read_sample_id, dtype=np.int32)
        self.individual_id = np.ascontiguousarray(individual_id, dtype=np.uint32)
        self.triple_ids = np.ascontiguousarray(triple_ids, dtype=np.uint32).reshape(-1)
        n_ind = self.individual_id.size
        genotype = np.ascontiguousarray(genotype, dtype=np.uint8)
        if n_variants is None:
            n_variants = genotype.size // n_ind if n_ind else 0
        self.n_variants = int(n_variants)
        self.genotype = genotype.reshape(-1)
        assert self.genotype.size == n_ind * self.n_variants
        if genotype_likelihoods is None:
            self.genotype_likelihoods = None
        else:
            self.genotype_likelihoods = np.ascontiguousarray(genotype_likelihoods, dtype=np.float64).reshape(-1)
            assert self.genotype_likelihoods.size == n_ind * self.n_variants * 3
        self.recombcost = np.ascontiguousarray(recombcost, dtype=np.uint32)
        self.positions = None if positions is None else np.ascontiguousarray(positions, dtype=np.uint32)
        self.distrust_genotypes = bool(distrust_genotypes)
        self.n_reads = self.read_sample_id.size
        assert self.read_ptr.size == self.n_reads + 1

        self.readset_view = ReadSetView(
            self.n_reads,
            _ptr(self.read_ptr, C.c_uint64),
            _ptr(self.var_position, C.c_int32),
            _ptr(self.var_allele, C.c_uint8),
            _ptr(self.var_quality, C.c_uint32),
            _ptr(self.read_sample_id, C.c_int32),
        )
        self.pedigree_view = PedigreeView(
            n_ind,
            _ptr(self.individual_id, C.c_uint32),
            self.triple_ids.size // 3,
            _ptr(self.triple_ids, C.c_uint32),
            self.n_variants,
            _ptr(self.genotype, C.c_uint8),
            _ptr(self.genotype_likelihoods, C.c_double),
            _ptr(None, C.c_uint8),
        )

    @property
    def n_individuals(self) -> int:
        return int(self.individual_id.size)

    def call_args(self):
        """(readset*, recombcost*, n_recomb, pedigree*, distrust, positions*, n_positions)"""
        return (
            C.byref(self.readset_view),
            _ptr(self.recombcost, C.c_uint32),
            C.c_size_t(self.recombcost.size),
            C.byref(self.pedigree_view),
            C.c_int(1 if self.distrust_genotypes else 0),
            _ptr(self.positions, C.c_uint32),
            C.c_size_t(0 if self.positions is None else self.positions.size),
        )


_lib = None
_debug_lib = None
_load_lock = threading.Lock()   # (the first call may come from several threads at once -- blocks.solve_blocks' create workers: ONE CDLL object per library)


def lib() -> C.CDLL:
    """Loads libwhatshap_amd.so (once).  Raises if it is missing -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C whatshap_amd/csrc). "
                "whatshap_amd has no CPU fallback."
            )
        _lib = _bind(C.CDLL(LIB_PATH), LIB_PATH)
    return _lib


def debug_lib() -> C.CDLL:
    """TEST INFRASTRUCTURE: libwhatshap_amd_debug.so -- the same sources compiled with -DWHAMD_DEBUG_BUILD: the CPU plan emulators, the
    host instantiation of the heuristic (include/whatshap_amd_debug.h), the kernel instantiations with cycle stamps and the timing switches
    (WHAMD_SLOT_STAMPS, WHAMD_SLOT_SKIP: results invalid).  None of that is in the product library.  A handle made by one library is only
    ever passed to functions of the same library."""
    global _debug_lib
    if _debug_lib is not None:
        return _debug_lib
    if not os.path.exists(DEBUG_LIB_PATH):
        raise ImportError(f"{DEBUG_LIB_PATH} is missing: make -C whatshap_amd/csrc debug (tests and timing scripts only)")
    L = _bind(C.CDLL(DEBUG_LIB_PATH), DEBUG_LIB_PATH)
    L.whamd_debug_emulate_slot_plan.restype = C.c_int
    L.whamd_debug_emulate_slot_plan.argtypes = [
        C.POINTER(ReadSetView), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(PedigreeView), C.c_int,
        C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
    ]
    L.whamd_debug_emulate_pedslot_plan.restype = C.c_int
    L.whamd_debug_emulate_pedslot_plan.argtypes = [
        C.POINTER(ReadSetView), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(PedigreeView), C.c_int,
        C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
    ]
    L.whamd_debug_pedmec_heuristic_create_host.restype = C.c_int
    L.whamd_debug_pedmec_heuristic_create_host.argtypes = [C.POINTER(ReadSetView), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(PedigreeView), C.c_int,
                                                           C.POINTER(C.c_uint32), C.c_size_t, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
    _debug_lib = L
    return L


def use_debug_library() -> None:
    """Timing scripts (scripts/gpu_slot_stamps.py ...): every later call of this process goes through the debug build."""
    global _lib
    _lib = debug_lib()


def _bind(L: C.CDLL, path: str) -> C.CDLL:
    H = C.c_void_p
    L.whamd_abi_version.restype = C.c_int
    L.whamd_device_count.restype = C.c_int
    L.whamd_last_error.restype = C.c_char_p
    L.whamd_dptable_create.restype = C.c_int
    L.whamd_dptable_create.argtypes = [
        C.POINTER(ReadSetView), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(PedigreeView), C.c_int,
        C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.POINTER(H),
    ]
    L.whamd_dptable_create_with_options.restype = C.c_int
    L.whamd_dptable_create_with_options.argtypes = [
        C.POINTER(ReadSetView), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(PedigreeView), C.c_int,
        C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_size_t, C.c_int, C.POINTER(H),
    ]
    L.whamd_dptable_solve.restype = C.c_int
    L.whamd_dptable_solve.argtypes = [H]
    L.whamd_dptable_enqueue.restype = C.c_int
    L.whamd_dptable_enqueue.argtypes = [H]
    L.whamd_dptable_wait.restype = C.c_int
    L.whamd_dptable_wait.argtypes = [H]
    L.whamd_dptable_enqueue_many.restype = C.c_int
    L.whamd_dptable_enqueue_many.argtypes = [C.POINTER(H), C.c_size_t]
    L.whamd_dptable_wait_many.restype = C.c_int
    L.whamd_dptable_wait_many.argtypes = [C.POINTER(H), C.c_size_t]
    L.whamd_dptable_release_device.restype = C.c_int
    L.whamd_dptable_release_device.argtypes = [H]
    L.whamd_dptable_destroy.restype = None
    L.whamd_dptable_destroy.argtypes = [H]
    L.whamd_dptable_column_count.restype = C.c_uint64
    L.whamd_dptable_column_count.argtypes = [H]
    L.whamd_dptable_individual_count.restype = C.c_uint32
    L.whamd_dptable_individual_count.argtypes = [H]
    L.whamd_dptable_read_count.restype = C.c_uint32
    L.whamd_dptable_read_count.argtypes = [H]
    L.whamd_dptable_positions.restype = C.c_int
    L.whamd_dptable_positions.argtypes = [H, C.POINTER(C.c_uint32)]
    L.whamd_dptable_get_optimal_score.restype = C.c_int
    L.whamd_dptable_get_optimal_score.argtypes = [H, C.POINTER(C.c_uint32)]
    L.whamd_dptable_get_super_reads.restype = C.c_int
    L.whamd_dptable_get_super_reads.argtypes = [
        H, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
    ]
    L.whamd_dptable_get_optimal_partitioning.restype = C.c_int
    L.whamd_dptable_get_optimal_partitioning.argtypes = [H, C.POINTER(C.c_uint8)]
    L.whamd_dptable_get_index_path.restype = C.c_int
    L.whamd_dptable_get_index_path.argtypes = [H, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.whamd_dptable_get_stats.restype = C.c_int
    L.whamd_dptable_get_stats.argtypes = [H, C.POINTER(SolveStats)]
    L.whamd_dptable_set_option.restype = C.c_int
    L.whamd_dptable_set_option.argtypes = [H, C.c_char_p, C.c_char_p]
    L.whamd_plan_summarize.restype = C.c_int
    L.whamd_plan_summarize.argtypes = [
        C.POINTER(ReadSetView), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(PedigreeView), C.c_int,
        C.POINTER(C.c_uint32), C.c_size_t, C.c_char_p, C.POINTER(PlanSummary),
    ]
    heur_args = [C.POINTER(ReadSetView), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(PedigreeView), C.c_int,
                 C.POINTER(C.c_uint32), C.c_size_t, C.c_uint32, C.c_int]
    L.whamd_pedmec_heuristic_create.restype = C.c_int
    L.whamd_pedmec_heuristic_create.argtypes = heur_args + [C.c_int, C.POINTER(C.c_void_p)]
    L.whamd_pedmec_heuristic_enqueue_many.restype = C.c_int
    L.whamd_pedmec_heuristic_enqueue_many.argtypes = [C.POINTER(HeuristicJob), C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    L.whamd_pedmec_heuristic_wait.restype = C.c_int
    L.whamd_pedmec_heuristic_wait.argtypes = [C.c_void_p]
    L.whamd_pedmec_heuristic_column_count.restype = C.c_uint64
    L.whamd_pedmec_heuristic_column_count.argtypes = [C.c_void_p]
    L.whamd_pedmec_heuristic_sample_count.restype = C.c_uint32
    L.whamd_pedmec_heuristic_sample_count.argtypes = [C.c_void_p]
    L.whamd_pedmec_heuristic_read_count.restype = C.c_uint32
    L.whamd_pedmec_heuristic_read_count.argtypes = [C.c_void_p]
    L.whamd_pedmec_heuristic_get.restype = C.c_int
    L.whamd_pedmec_heuristic_get.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_int8),
                                             C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.whamd_pedmec_heuristic_get_stats.restype = C.c_int
    L.whamd_pedmec_heuristic_get_stats.argtypes = [C.c_void_p, C.POINTER(HeuristicStats)]
    L.whamd_pedmec_heuristic_destroy.restype = None
    L.whamd_pedmec_heuristic_destroy.argtypes = [C.c_void_p]
    L.whamd_read_sort_hash.restype = C.c_uint64
    L.whamd_read_sort_hash.argtypes = [C.c_char_p, C.c_int]
    L.whamd_genotype_likelihoods.restype = C.c_int
    L.whamd_genotype_likelihoods.argtypes = [
        C.POINTER(ReadSetView), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(PedigreeView), C.POINTER(C.c_uint32), C.c_size_t,
        C.c_int, C.c_uint32, C.POINTER(C.c_double), C.c_size_t, C.POINTER(GenotypeStats),
    ]
    L.whamd_readselection.restype = C.c_int
    L.whamd_readselection.argtypes = [
        C.POINTER(ReadSetView), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_size_t, C.c_uint32, C.c_int,
        C.POINTER(C.c_uint8), C.POINTER(C.c_uint64),
    ]
    if L.whamd_abi_version() != ABI_VERSION:
        raise ImportError(f"{path} has ABI version {L.whamd_abi_version()}, this binding was written for {ABI_VERSION} (stale library? run make)")
    return L


# every symbol include/whatshap_amd.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = [
    "whamd_abi_version", "whamd_device_count", "whamd_device_pci_bus_id", "whamd_last_error", "whamd_dptable_create", "whamd_dptable_create_with_options", "whamd_dptable_solve",
    "whamd_dptable_release_device", "whamd_dptable_destroy", "whamd_dptable_column_count", "whamd_dptable_individual_count",
    "whamd_dptable_read_count", "whamd_dptable_positions", "whamd_dptable_get_optimal_score",
    "whamd_dptable_get_super_reads", "whamd_dptable_get_optimal_partitioning", "whamd_dptable_get_index_path",
    "whamd_dptable_get_stats", "whamd_dptable_set_option", "whamd_read_sort_hash", "whamd_plan_summarize",
    "whamd_dptable_enqueue", "whamd_dptable_wait", "whamd_dptable_enqueue_many", "whamd_dptable_wait_many",
    "whamd_pedmec_heuristic_create",
    "whamd_pedmec_heuristic_enqueue_many", "whamd_pedmec_heuristic_wait",
    "whamd_pedmec_heuristic_column_count", "whamd_pedmec_heuristic_sample_count", "whamd_pedmec_heuristic_read_count",
    "whamd_pedmec_heuristic_get", "whamd_pedmec_heuristic_get_stats", "whamd_pedmec_heuristic_destroy",
    "whamd_readselection", "whamd_genotype_likelihoods", "whamd_release_caches", "whamd_host_pool_idle_bytes",
]


class SolverError(RuntimeError):
    """RuntimeError with the library's status code attached (the reference raises plain RuntimeError)."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


def _check(status: int, L=None):
    if status != WHAMD_OK:
        raise SolverError(status, (L or lib()).whamd_last_error().decode("utf-8", "replace"))


def _library_of(tables):
    """The library that created these tables (product or debug build: never mixed in one call)."""
    libs = {id(getattr(t, "_L", None)): getattr(t, "_L", None) for t in tables}
    if len(libs) != 1:
        raise ValueError("tables of the product and of the debug library cannot share a call")
    return next(iter(libs.values())) or lib()


def enqueue_many(tables) -> None:
    """whamd_dptable_enqueue_many: submits the solves of several tables with interleaved launch sequences."""
    tables = list(tables)
    if not tables:
        return
    arr = (C.c_void_p * len(tables))(*[t._h.value for t in tables])
    _check(_library_of(tables).whamd_dptable_enqueue_many(arr, len(tables)))


def wait_many(tables) -> None:
    """whamd_dptable_wait_many: collects several tables in flight; their host-side result extraction runs on a few threads at once."""
    tables = list(tables)
    if not tables:
        return
    arr = (C.c_void_p * len(tables))(*[t._h.value for t in tables])
    _check(_library_of(tables).whamd_dptable_wait_many(arr, len(tables)))


class NativeTable:
    """Thin RAII wrapper of whamd_dptable: create -> (set_option) -> solve -> getters."""

    def __init__(self, problem: ProblemArrays, device: int = 0, path: Optional[str] = None, solve: bool = True, options: Optional[dict] = None):
        L = self._L = lib()   # the library that makes the handle also queries and destroys it (the debug build has pools and arenas of its own)
        self._h = C.c_void_p()
        self._problem = problem  # keep the arrays alive while create() reads them
        opts = dict(options or {})
        if path is not None:
            opts["path"] = path
        if opts:   # applied before the plan is made: one upload
            keys = (C.c_char_p * len(opts))(*[str(k).encode() for k in opts])
            values = (C.c_char_p * len(opts))(*[str(v).encode() for v in opts.values()])
            _check(L.whamd_dptable_create_with_options(*problem.call_args(), keys, values, C.c_size_t(len(opts)), C.c_int(device), C.byref(self._h)))
        else:
            _check(L.whamd_dptable_create(*problem.call_args(), C.c_int(device), C.byref(self._h)))
        self.n_columns = int(L.whamd_dptable_column_count(self._h))
        self.n_individuals = int(L.whamd_dptable_individual_count(self._h))
        self.n_reads = int(L.whamd_dptable_read_count(self._h))
        if solve:
            self.solve()

    def set_option(self, key: str, value: str):
        _check(self._L.whamd_dptable_set_option(self._h, key.encode(), value.encode()))

    def solve(self):
        _check(self._L.whamd_dptable_solve(self._h))

    def enqueue(self):
        """Submit the solve to the table's stream without waiting (pair with wait())."""
        _check(self._L.whamd_dptable_enqueue(self._h))

    def wait(self):
        _check(self._L.whamd_dptable_wait(self._h))

    def release_device(self):
        """Frees the device side of a solved table; the getters keep working."""
        _check(self._L.whamd_dptable_release_device(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.whamd_dptable_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def positions(self) -> np.ndarray:
        out = np.zeros(self.n_columns, dtype=np.uint32)
        _check(self._L.whamd_dptable_positions(self._h, _ptr(out, C.c_uint32)))
        return out

    def optimal_score(self) -> int:
        v = C.c_uint32()
        _check(self._L.whamd_dptable_get_optimal_score(self._h, C.byref(v)))
        return int(v.value)

    def super_reads(self):
        n, ni = self.n_columns, self.n_individuals
        a0 = np.zeros((ni, n), dtype=np.uint8)
        a1 = np.zeros((ni, n), dtype=np.uint8)
        q = np.zeros((ni, n), dtype=np.uint32)
        tv = np.zeros(n, dtype=np.uint32)
        sid = np.zeros(ni, dtype=np.uint32)
        _check(self._L.whamd_dptable_get_super_reads(
            self._h, _ptr(a0, C.c_uint8), _ptr(a1, C.c_uint8), _ptr(q, C.c_uint32), _ptr(tv, C.c_uint32), _ptr(sid, C.c_uint32)))
        return a0, a1, q, tv, sid

    def partitioning(self) -> np.ndarray:
        out = np.zeros(self.n_reads, dtype=np.uint8)
        _check(self._L.whamd_dptable_get_optimal_partitioning(self._h, _ptr(out, C.c_uint8)))
        return out

    def index_path(self):
        idx = np.zeros(self.n_columns, dtype=np.uint32)
        tv = np.zeros(self.n_columns, dtype=np.uint32)
        _check(self._L.whamd_dptable_get_index_path(self._h, _ptr(idx, C.c_uint32), _ptr(tv, C.c_uint32)))
        return idx, tv

    def stats(self) -> dict:
        s = SolveStats()
        _check(self._L.whamd_dptable_get_stats(self._h, C.byref(s)))
        return s.as_dict()


def plan_summary(problem: ProblemArrays, path: str = "auto") -> dict:
    """Host-only: how whamd_dptable_create would schedule the columns of `problem` (no device needed)."""
    out = PlanSummary()
    _check(lib().whamd_plan_summarize(*problem.call_args(), path.encode(), C.byref(out)))
    return out.as_dict()


def emulate_slot_plan(problem: ProblemArrays, n_columns: int, slot_l: int = 11, symmetry: int = 1, slot_r: int = 2):
    """Host-only planner diagnostic (whamd_debug_emulate_slot_plan): (index path, optimal score, columns inside runs).
    slot_r: reg slots per thread (1, 2 or 3; passed as slot_l + 100 for 3, + 200 for 1)."""
    if slot_r >= 3:
        slot_l += 100
    elif slot_r <= 1:
        slot_l += 200
    idx = np.zeros(max(n_columns, 1), dtype=np.uint32)
    score = C.c_uint32()
    ncols = C.c_uint64()
    D = debug_lib()
    _check(D.whamd_debug_emulate_slot_plan(*problem.call_args(), C.c_int(slot_l), C.c_int(symmetry), _ptr(idx, C.c_uint32),
                                           C.byref(score), C.byref(ncols)), D)
    return idx[:n_columns], int(score.value), int(ncols.value)


def emulate_pedslot_plan(problem: ProblemArrays, n_columns: int, slot_l: int = 0):
    """Host-only diagnostic of the pedigree slot plan (whamd_debug_emulate_pedslot_plan):
    (index path, transmission path, optimal score, columns inside runs)."""
    idx = np.zeros(max(n_columns, 1), dtype=np.uint32)
    trans = np.zeros(max(n_columns, 1), dtype=np.uint32)
    score = C.c_uint32()
    ncols = C.c_uint64()
    D = debug_lib()
    _check(D.whamd_debug_emulate_pedslot_plan(*problem.call_args(), C.c_int(slot_l), _ptr(idx, C.c_uint32), _ptr(trans, C.c_uint32),
                                              C.byref(score), C.byref(ncols)), D)
    return idx[:n_columns], trans[:n_columns], int(score.value), int(ncols.value)


def debug_lazy_terms_check(problem: ProblemArrays, need=None, rounds: int = 1) -> dict:
    """whamd_debug_lazy_terms_check: the lazily built generic term lists (+ fill_lazy_terms on the columns of `need`, in `rounds` calls) against the eager ones."""
    D = debug_lib()
    fn = D.whamd_debug_lazy_terms_check
    fn.restype = C.c_int
    lazy, diff, before, after = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    need_arr = None if need is None else np.ascontiguousarray(need, dtype=np.uint8)
    _check(fn(*problem.call_args(), None if need_arr is None else _ptr(need_arr, C.c_uint8), C.c_int(int(rounds)), C.byref(lazy), C.byref(diff),
              C.byref(before), C.byref(after)), D)
    return {"lazy": bool(lazy.value), "differences": int(diff.value), "built_before": int(before.value), "built_after": int(after.value)}


def _heuristic_result(L, h) -> dict:
    n = int(L.whamd_pedmec_heuristic_column_count(h))
    ns = int(L.whamd_pedmec_heuristic_sample_count(h))
    nr = int(L.whamd_pedmec_heuristic_read_count(h))
    score = C.c_float()
    bip = np.zeros(max(nr, 1), dtype=np.uint8)
    trans = np.zeros(max(n, 1), dtype=np.uint32)
    haps = np.zeros((max(ns, 1), 2, max(n, 1)), dtype=np.int8)
    mut = np.zeros((max(ns, 1), 2, max(n, 1)), dtype=np.uint8)
    sid = np.zeros(max(ns, 1), dtype=np.uint32)
    pos = np.zeros(max(n, 1), dtype=np.uint32)
    _check(L.whamd_pedmec_heuristic_get(h, C.byref(score), _ptr(bip, C.c_uint8), _ptr(trans, C.c_uint32), haps.ctypes.data_as(C.POINTER(C.c_int8)),
                                        _ptr(mut, C.c_uint8), _ptr(sid, C.c_uint32), _ptr(pos, C.c_uint32)), L)
    stats = HeuristicStats()
    _check(L.whamd_pedmec_heuristic_get_stats(h, C.byref(stats)), L)
    return {"score": float(score.value), "bipartition": bip[:nr], "transmission": trans[:n], "haplotypes": haps[:ns, :, :n], "mutated": mut[:ns, :, :n],
            "sample_ids": sid[:ns], "positions": pos[:n], "stats": stats.as_dict()}


def pedmec_heuristic(problem: ProblemArrays, row_limit: int = 256, allow_mutations: bool = True, device: int = 0, host_diagnostic: bool = False) -> dict:
    """whamd_pedmec_heuristic_create + _get: the beam search of PedMecHeuristic (constructor + solve) and everything its getters
    return.  host_diagnostic: the same solver source on one CPU thread (tests only)."""
    L = debug_lib() if host_diagnostic else lib()   # (the handle stays with the library that made it)
    h = C.c_void_p()
    a = problem.call_args()
    if host_diagnostic:
        _check(L.whamd_debug_pedmec_heuristic_create_host(*a, C.c_uint32(int(row_limit)), C.c_int(1 if allow_mutations else 0), C.byref(h)), L)
    else:
        _check(L.whamd_pedmec_heuristic_create(*a, C.c_uint32(int(row_limit)), C.c_int(1 if allow_mutations else 0), C.c_int(int(device)), C.byref(h)))
    try:
        return _heuristic_result(L, h)
    finally:
        L.whamd_pedmec_heuristic_destroy(h)


class HeuristicBatch:
    """whamd_pedmec_heuristic_enqueue_many: several tables in ONE launch (one persistent workgroup each) on a stream of the batch's own;
    `results()` waits and returns one dict per table (as pedmec_heuristic)."""

    def __init__(self, problems, row_limit: int = 256, allow_mutations: bool = True, device: int = 0):
        L = lib()
        problems = list(problems)
        self._L = L
        self._n = len(problems)
        self._handles = (C.c_void_p * max(self._n, 1))()
        jobs = (HeuristicJob * max(self._n, 1))()
        for i, p in enumerate(problems):   # (the problems outlive the call: the arrays are only read during _enqueue_many)
            jobs[i].readset = C.pointer(p.readset_view)
            jobs[i].recombcost = _ptr(p.recombcost, C.c_uint32)
            jobs[i].n_recombcost = p.recombcost.size
            jobs[i].pedigree = C.pointer(p.pedigree_view)
            jobs[i].distrust_genotypes = 1 if p.distrust_genotypes else 0
            jobs[i].positions = _ptr(p.positions, C.c_uint32)
            jobs[i].n_positions = 0 if p.positions is None else p.positions.size
            jobs[i].row_limit = int(row_limit)
            jobs[i].allow_mutations = 1 if allow_mutations else 0
        if self._n:
            _check(L.whamd_pedmec_heuristic_enqueue_many(jobs, self._n, C.c_int(int(device)), self._handles))

    def results(self):
        out = []
        try:
            for i in range(self._n):
                _check(self._L.whamd_pedmec_heuristic_wait(self._handles[i]))
                out.append(_heuristic_result(self._L, self._handles[i]))
        finally:
            self.close()
        return out

    def close(self):
        for i in range(self._n):
            if self._handles[i]:
                self._L.whamd_pedmec_heuristic_destroy(self._handles[i])
                self._handles[i] = None
        self._n = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pedmec_heuristic_many(problems, row_limit: int = 256, allow_mutations: bool = True, device: int = 0):
    """Several tables through one batched launch; one result dict per table."""
    return HeuristicBatch(problems, row_limit, allow_mutations, device).results()


def host_pool_idle_bytes() -> int:
    """whamd_host_pool_idle_bytes: host memory the library keeps between tables (given back by release_caches())."""
    fn = lib().whamd_host_pool_idle_bytes
    fn.restype = C.c_uint64
    return int(fn())


def device_count() -> int:
    return int(lib().whamd_device_count())


def device_pci_bus_id(device: int) -> str:
    """whamd_device_pci_bus_id: "0000:c5:00.0" of HIP device `device` (raises SolverError / WHAMD_ERR_DEVICE if there is none)."""
    buf = C.create_string_buffer(64)
    fn = lib().whamd_device_pci_bus_id
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    _check(fn(int(device), buf, 64))
    return buf.value.decode()


def readselection(read_ptr, var_position, var_quality, max_cov: int, read_source_id=None, preferred_source_ids=None,
                  bridging: bool = True) -> np.ndarray:
    """whamd_readselection on flat arrays: boolean mask [n_reads] of the selected reads."""
    read_ptr = np.ascontiguousarray(read_ptr, dtype=np.uint64)
    if read_ptr.size == 0:
        read_ptr = np.zeros(1, dtype=np.uint64)
    var_position = np.ascontiguousarray(var_position, dtype=np.int32)
    var_quality = np.ascontiguousarray(var_quality, dtype=np.uint32)
    n_reads = read_ptr.size - 1
    assert var_position.size == var_quality.size == int(read_ptr[-1])
    view = ReadSetView(n_reads, _ptr(read_ptr, C.c_uint64), _ptr(var_position, C.c_int32), None, _ptr(var_quality, C.c_uint32), None)
    preferred = np.ascontiguousarray(sorted(preferred_source_ids) if preferred_source_ids is not None else [], dtype=np.int32)
    sources = None
    if preferred.size:
        sources = np.ascontiguousarray(read_source_id, dtype=np.int32)
        assert sources.size == n_reads
    selected = np.zeros(max(n_reads, 1), dtype=np.uint8)
    count = C.c_uint64()
    _check(lib().whamd_readselection(C.byref(view), None if sources is None else _ptr(sources, C.c_int32),
                                     _ptr(preferred, C.c_int32) if preferred.size else None, preferred.size,
                                     C.c_uint32(int(max_cov)), C.c_int(1 if bridging else 0), _ptr(selected, C.c_uint8), C.byref(count)))
    return selected[:n_reads].astype(bool)


def genotype_likelihoods(problem: ProblemArrays, n_columns: int, device: int = 0, window: int = 0):
    """whamd_genotype_likelihoods: (likelihoods [individuals, columns, 3], stats dict)."""
    n_ind = problem.n_individuals
    gl = np.zeros((n_ind, int(n_columns), 3), dtype=np.float64)
    stats = GenotypeStats()
    a = problem.call_args()
    _check(lib().whamd_genotype_likelihoods(a[0], a[1], a[2], a[3], a[5], a[6], C.c_int(int(device)), C.c_uint32(int(window)),
                                            _ptr(gl, C.c_double), C.c_size_t(max(gl.size, 1) if gl.size else 0), C.byref(stats)))
    if int(stats.n_columns) != int(n_columns):
        # the library strides its output by ITS column count: a different count here would silently mis-align every individual after the first
        raise SolverError(WHAMD_ERR_INVALID, f"genotype_likelihoods: caller assumed {n_columns} columns, the library found {int(stats.n_columns)}")
    return gl, stats.as_dict()


def release_caches() -> None:
    """whamd_release_caches: frees the device memory the genotyping path keeps between calls."""
    lib().whamd_release_caches.restype = None
    lib().whamd_release_caches()


def read_sort_hash(name: str, source_id: int) -> int:
    return int(lib().whamd_read_sort_hash(name.encode("utf-8"), int(source_id)))
