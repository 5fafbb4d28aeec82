"""TEST INFRASTRUCTURE -- not product code.

ctypes wrappers of the two CPU checkers:

* ``OracleTable``    : oracle/pedmec_oracle.c, the plain-C restatement of the reference algorithm
  (built into oracle/_build/libpedmec_oracle.so by oracle/Makefile).
* ``ReferenceTable`` : oracle/_ref/libwhatshap_ref.so, the REAL reference C++ compiled from
  /root/reference/src plus oracle/ref_driver.cpp (only present where it was built; the .so travels
  to the GPU box, the sources do not).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Both tables take the same ``whatshap_amd._native.ProblemArrays`` the product library takes.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from whatshap_amd._native import PedigreeView, ProblemArrays, ReadSetView, _ptr

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "_build", "libpedmec_oracle.so")
REFERENCE_LIB = os.path.join(_HERE, "_ref", "libwhatshap_ref.so")


def build(reference: str = "/root/reference") -> None:
    """Compiles the C restatement and, when the reference tree is present, oracle/_ref."""
    subprocess.run(["make", "-C", _HERE, "oracle"], check=True, stdout=subprocess.DEVNULL)
    if os.path.isdir(os.path.join(reference, "src")):
        subprocess.run(["make", "-C", _HERE, "ref", f"REFERENCE={reference}"], check=True, stdout=subprocess.DEVNULL)
        # the reference's Python extension modules (whatshap.core, readselect) for the class-switch tests; optional
        try:
            from . import build_cython_ref

            build_cython_ref.build()
        except Exception as exc:  # a missing cython / compiler problem must not take the C oracle down with it
            print(f"oracle: reference extension modules not built ({exc})")


def have_reference() -> bool:
    return os.path.exists(REFERENCE_LIB)


class OracleError(RuntimeError):
    pass


class _Table:
    _prefix = ""
    _libpath = ""
    _lib = None

    @classmethod
    def _load(cls):
        if cls._lib is None:
            if not os.path.exists(cls._libpath):
                if cls is OracleTable:
                    build()
                if not os.path.exists(cls._libpath):
                    raise OracleError(f"{cls._libpath} is not built (make -C oracle)")
            L = C.CDLL(cls._libpath)
            p = cls._prefix
            H = C.c_void_p
            getattr(L, p + "create").restype = C.c_int
            getattr(L, p + "create").argtypes = [
                C.POINTER(ReadSetView), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(PedigreeView), C.c_int,
                C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(H),
            ]
            getattr(L, p + "error").restype = C.c_char_p
            getattr(L, p + "error").argtypes = [H]
            getattr(L, p + "column_count").restype = C.c_uint32
            getattr(L, p + "column_count").argtypes = [H]
            getattr(L, p + "optimal_score").restype = C.c_uint32
            getattr(L, p + "optimal_score").argtypes = [H]
            getattr(L, p + "positions").restype = None
            getattr(L, p + "positions").argtypes = [H, C.POINTER(C.c_uint32)]
            getattr(L, p + "index_path").restype = None
            getattr(L, p + "index_path").argtypes = [H, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
            getattr(L, p + "super_reads").restype = C.c_int
            getattr(L, p + "super_reads").argtypes = [
                H, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
            getattr(L, p + "partitioning").restype = None
            getattr(L, p + "partitioning").argtypes = [H, C.POINTER(C.c_uint8)]
            getattr(L, p + "destroy").restype = None
            getattr(L, p + "destroy").argtypes = [H]
            cls._configure(L)
            cls._lib = L
        return cls._lib

    @classmethod
    def _configure(cls, L):
        pass

    def _fn(self, name):
        return getattr(self._load(), self._prefix + name)

    def __init__(self, problem: ProblemArrays):
        self._h = C.c_void_p()
        self._problem = problem
        status = self._fn("create")(*problem.call_args(), C.byref(self._h))
        if status != 0:
            message = self._fn("error")(self._h).decode("utf-8", "replace")
            self.close()
            raise OracleError(message)
        self.n_columns = int(self._fn("column_count")(self._h))
        self.n_individuals = problem.n_individuals
        self.n_reads = problem.n_reads

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def positions(self):
        out = np.zeros(self.n_columns, dtype=np.uint32)
        self._fn("positions")(self._h, _ptr(out, C.c_uint32))
        return out

    def optimal_score(self) -> int:
        return int(self._fn("optimal_score")(self._h))

    def index_path(self):
        idx = np.zeros(self.n_columns, dtype=np.uint32)
        tv = np.zeros(self.n_columns, dtype=np.uint32)
        self._fn("index_path")(self._h, _ptr(idx, C.c_uint32), _ptr(tv, C.c_uint32))
        return idx, tv

    def super_reads(self):
        n, ni = self.n_columns, self.n_individuals
        a0 = np.zeros((ni, n), dtype=np.uint8)
        a1 = np.zeros((ni, n), dtype=np.uint8)
        q = np.zeros((ni, n), dtype=np.uint32)
        tv = np.zeros(n, dtype=np.uint32)
        sid = np.zeros(ni, dtype=np.uint32)
        status = self._fn("super_reads")(self._h, _ptr(a0, C.c_uint8), _ptr(a1, C.c_uint8), _ptr(q, C.c_uint32),
                                         _ptr(tv, C.c_uint32), _ptr(sid, C.c_uint32))
        if status != 0:
            raise OracleError(self._fn("error")(self._h).decode("utf-8", "replace"))
        return a0, a1, q, tv, sid

    def partitioning(self):
        out = np.zeros(self.n_reads, dtype=np.uint8)
        self._fn("partitioning")(self._h, _ptr(out, C.c_uint8))
        return out


class OracleTable(_Table):
    """The plain-C restatement (oracle/pedmec_oracle.c)."""

    _prefix = "pmo_"
    _libpath = ORACLE_LIB
    _lib = None

    @classmethod
    def _configure(cls, L):
        L.pmo_cell_count.restype = C.c_uint64
        L.pmo_cell_count.argtypes = [C.c_void_p]

    def cell_count(self) -> int:
        return int(self._load().pmo_cell_count(self._h))


class ReferenceTable(_Table):
    """The compiled reference (src/pedigreedptable.cpp et al. + oracle/ref_driver.cpp)."""

    _prefix = "whref_"
    _libpath = REFERENCE_LIB
    _lib = None

    @classmethod
    def _configure(cls, L):
        L.whref_ctor_seconds.restype = C.c_double
        L.whref_ctor_seconds.argtypes = [C.c_void_p]

    def ctor_seconds(self) -> float:
        return float(self._load().whref_ctor_seconds(self._h))


class ReferenceHeuristic:
    """The compiled reference's PedMecHeuristic (src/pedmecheuristic.cpp + oracle/ref_driver.cpp), solved at construction.
    ``problem`` as for the tables (reads sorted, individuals in ascending sample-id order)."""

    _lib = None

    @classmethod
    def _load(cls):
        if cls._lib is None:
            if not os.path.exists(REFERENCE_LIB):
                raise OracleError(f"{REFERENCE_LIB} is not built (make -C oracle)")
            L = C.CDLL(REFERENCE_LIB)
            H = C.c_void_p
            L.whref_heuristic_create.restype = C.c_int
            L.whref_heuristic_create.argtypes = [C.POINTER(ReadSetView), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(PedigreeView), C.c_int,
                                                 C.POINTER(C.c_uint32), C.c_size_t, C.c_uint32, C.c_int, C.POINTER(H)]
            L.whref_heuristic_error.restype = C.c_char_p
            L.whref_heuristic_error.argtypes = [H]
            for name in ("column_count", "sample_count"):
                getattr(L, "whref_heuristic_" + name).restype = C.c_uint32
                getattr(L, "whref_heuristic_" + name).argtypes = [H]
            L.whref_heuristic_solve_seconds.restype = C.c_double
            L.whref_heuristic_solve_seconds.argtypes = [H]
            L.whref_heuristic_score.restype = C.c_float
            L.whref_heuristic_score.argtypes = [H]
            L.whref_heuristic_bipartition.restype = None
            L.whref_heuristic_bipartition.argtypes = [H, C.POINTER(C.c_uint8)]
            L.whref_heuristic_transmission.restype = None
            L.whref_heuristic_transmission.argtypes = [H, C.POINTER(C.c_uint32)]
            L.whref_heuristic_haplotypes.restype = None
            L.whref_heuristic_haplotypes.argtypes = [H, C.POINTER(C.c_int8), C.POINTER(C.c_uint8)]
            L.whref_heuristic_destroy.restype = None
            L.whref_heuristic_destroy.argtypes = [H]
            cls._lib = L
        return cls._lib

    def __init__(self, problem: ProblemArrays, row_limit: int = 256, allow_mutations: bool = True):
        L = self._load()
        self._h = C.c_void_p()
        self._problem = problem
        status = L.whref_heuristic_create(*problem.call_args(), C.c_uint32(int(row_limit)), C.c_int(1 if allow_mutations else 0), C.byref(self._h))
        if status != 0:
            message = L.whref_heuristic_error(self._h).decode("utf-8", "replace")
            self.close()
            raise OracleError(message)
        self.n_columns = int(L.whref_heuristic_column_count(self._h))
        self.n_samples = int(L.whref_heuristic_sample_count(self._h))
        self.n_reads = problem.n_reads

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._load().whref_heuristic_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def score(self) -> float:
        return float(self._load().whref_heuristic_score(self._h))

    def solve_seconds(self) -> float:
        return float(self._load().whref_heuristic_solve_seconds(self._h))

    def bipartition(self):
        out = np.zeros(max(self.n_reads, 1), dtype=np.uint8)
        self._load().whref_heuristic_bipartition(self._h, _ptr(out, C.c_uint8))
        return out[:self.n_reads]

    def transmission(self):
        out = np.zeros(max(self.n_columns, 1), dtype=np.uint32)
        self._load().whref_heuristic_transmission(self._h, _ptr(out, C.c_uint32))
        return out[:self.n_columns]

    def haplotypes(self):
        """(alleles [samples][2][columns] int8, mutated [samples][2][columns] 0/1)"""
        haps = np.zeros((max(self.n_samples, 1), 2, max(self.n_columns, 1)), dtype=np.int8)
        mut = np.zeros((max(self.n_samples, 1), 2, max(self.n_columns, 1)), dtype=np.uint8)
        self._load().whref_heuristic_haplotypes(self._h, haps.ctypes.data_as(C.POINTER(C.c_int8)), _ptr(mut, C.c_uint8))
        return haps[:self.n_samples, :, :self.n_columns], mut[:self.n_samples, :, :self.n_columns]


def heuristic_tuple(h):
    """(score, bipartition, transmission, haplotypes, mutations) in comparable form."""
    haps, mut = h.haplotypes()
    return {"score": h.score(), "bipartition": h.bipartition().tolist(), "transmission": h.transmission().tolist(),
            "haplotypes": haps.tolist(), "mutated": mut.tolist()}


def solution_tuple(table):
    """(cost, index path, transmission vector, partitioning, superreads) in comparable form."""
    a0, a1, q, tv, sid = table.super_reads()
    idx, tv2 = table.index_path()
    return {
        "cost": table.optimal_score(),
        "index_path": idx.tolist(),
        "transmission": tv.tolist(),
        "path_transmission": tv2.tolist(),
        "partitioning": table.partitioning().tolist(),
        "allele0": a0.tolist(),
        "allele1": a1.tolist(),
        "quality": q.tolist(),
        "sample_ids": sid.tolist(),
        "positions": table.positions().tolist(),
    }
