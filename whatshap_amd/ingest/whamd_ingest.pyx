# cython: language_level=3
# distutils: language = c++
"""Compiled ingestion of WhatsHap's own objects: ``whatshap.core.ReadSet`` / ``Pedigree`` -> the flat arrays behind the
C-ABI views (``whatshap_amd._native.ProblemArrays``), walking the C++ objects through ``thisptr`` exactly as
``whatshap/readselect.pyx:14-15,244`` does.  Built against the reference's ``core.pxd`` / ``cpp.pxd`` / ``src/*.h`` by
``whatshap_amd/ingest/build.py`` (as any sibling extension of WhatsHap would be); import ``whatshap.core`` first
(``RTLD_GLOBAL``, as ``whatshap/__init__.py`` does) so that the C++ symbols resolve."""
from libc.stdint cimport int32_t, uint8_t, uint32_t, uint64_t
from libc.stdlib cimport free
from libcpp.vector cimport vector
from cython.view cimport array as cvarray

import numpy as np

from whatshap.core cimport Pedigree, ReadSet
from whatshap cimport cpp


cdef extern from "whamd_ingest_helpers.h":
    void* whamd_ingest_alloc(size_t)
    size_t whamd_readset_scan(cpp.ReadSet*, vector[cpp.Read*]&, uint64_t*) except +
    void whamd_flatten_readset(const vector[cpp.Read*]&, const uint64_t*, int32_t*, uint8_t*, uint32_t*, int32_t*) except +
    int whamd_flatten_pedigree(cpp.Pedigree*, uint32_t*, uint32_t*, uint8_t*, double*) except +
    void whamd_emit_superread_sets(size_t, int, const int32_t*, size_t, const uint32_t*, const uint8_t*, const uint8_t*, const uint32_t*, cpp.ReadSet**) except +
    void whamd_read_source_ids(cpp.ReadSet*, int32_t*) except +


cdef _uninitialised(size_t n, dtype):
    """1-d array of ``n`` items whose memory comes from ``whamd_ingest_alloc`` (huge pages for large arrays), freed with the array."""
    cdef size_t itemsize = np.dtype(dtype).itemsize
    cdef void* ptr = whamd_ingest_alloc(n * itemsize)
    if ptr == NULL:
        raise MemoryError()
    cdef cvarray owner = <char[:n * itemsize]> <char*> ptr
    owner.callback_free_data = free
    return np.frombuffer(owner, dtype=dtype, count=n)


def u32_array(values):
    """A Python sequence of non-negative ints (``recombcost``, ``positions`` as ``whatshap/cli/phase.py:604-612`` passes them: lists of
    200 000 ints) -> uint32 array, one typed loop instead of ``np.asarray(list(...))``'s generic conversion (7 ms per list)."""
    if isinstance(values, np.ndarray):
        return np.ascontiguousarray(values, dtype=np.uint32)
    if not isinstance(values, (list, tuple)):
        values = list(values)
    cdef size_t n = len(values), i
    out = np.empty(max(n, 1), dtype=np.uint32)
    cdef uint32_t[::1] v = out
    cdef object item
    for i in range(n):
        item = values[i]
        v[i] = <uint32_t>(<unsigned long>item)   # raises OverflowError for negative values, like vector[unsigned int] from a list
    return out[:n]


def flatten_readset(ReadSet readset):
    """(read_ptr, var_position, var_allele, var_quality, read_sample_id) of a reference ReadSet."""
    cdef size_t n_reads = readset.thisptr.size()
    cdef vector[cpp.Read*] reads
    read_ptr = np.zeros(n_reads + 1, dtype=np.uint64)
    cdef uint64_t[::1] v_ptr = read_ptr
    cdef size_t nnz = whamd_readset_scan(readset.thisptr, reads, &v_ptr[0])
    position = _uninitialised(max(nnz, 1), np.int32)
    allele = _uninitialised(max(nnz, 1), np.uint8)
    quality = _uninitialised(max(nnz, 1), np.uint32)
    sample = np.zeros(max(n_reads, 1), dtype=np.int32)
    cdef int32_t[::1] v_pos = position
    cdef uint8_t[::1] v_allele = allele
    cdef uint32_t[::1] v_quality = quality
    cdef int32_t[::1] v_sample = sample
    whamd_flatten_readset(reads, &v_ptr[0], &v_pos[0], &v_allele[0], &v_quality[0], &v_sample[0])
    return read_ptr, position[:nnz], allele[:nnz], quality[:nnz], sample[:n_reads]


def flatten_pedigree(Pedigree pedigree):
    """(individual ids, triple ids [3 * triples], genotype codes [individuals, variants], likelihoods [individuals,
    variants, 3] or None) of a reference Pedigree -- which is opaque from Python (no accessor for its individuals or trios)."""
    cdef size_t n_ind = pedigree.thisptr.size()
    cdef size_t n_var = pedigree.thisptr.get_variant_count() if n_ind else 0
    cdef size_t n_tri = pedigree.thisptr.triple_count()
    ids = np.zeros(max(n_ind, 1), dtype=np.uint32)
    triples = np.zeros(max(3 * n_tri, 1), dtype=np.uint32)
    genotype = np.zeros(max(n_ind * n_var, 1), dtype=np.uint8)
    gl = np.zeros(max(n_ind * n_var * 3, 1), dtype=np.float64)
    cdef uint32_t[::1] v_ids = ids
    cdef uint32_t[::1] v_triples = triples
    cdef uint8_t[::1] v_genotype = genotype
    cdef double[::1] v_gl = gl
    cdef int any_gl = whamd_flatten_pedigree(pedigree.thisptr, &v_ids[0], &v_triples[0], &v_genotype[0], &v_gl[0])
    return (ids[:n_ind], triples[:3 * n_tri], genotype[:n_ind * n_var].reshape(n_ind, n_var),
            gl[:n_ind * n_var * 3].reshape(n_ind, n_var, 3) if any_gl else None)


def read_source_ids(ReadSet readset):
    """``read.source_id`` of every read as one int32 array (``whatshap/readselect.pyx:52-56`` asks read by read)."""
    cdef size_t n_reads = readset.thisptr.size()
    out = np.zeros(max(n_reads, 1), dtype=np.int32)
    cdef int32_t[::1] v_out = out
    whamd_read_source_ids(readset.thisptr, &v_out[0])
    return out[:n_reads]


def emit_superreads(positions, allele0, allele1, quality, sample_ids, numbered=True):
    """The C-ABI arrays of ``whamd_dptable_get_super_reads`` (``allele0 / allele1`` u8 ``[individuals, columns]``, ``quality`` u32
    ``[individuals, columns]``, ``sample_ids`` ``[individuals]``) + the column positions -> one reference ``ReadSet`` per individual,
    built in C++ and adopted by a ``whatshap.core.ReadSet`` exactly as ``whatshap/core.pyx:388-400`` adopts what
    ``PedigreeDPTable::get_super_reads`` (``src/pedigreedptable.cpp:344-388``) filled: no Python object per variant.
    ``numbered=False``: the read names of ``PedMecHeuristic::getSuperReads`` (``src/pedmecheuristic.cpp:105-121``)."""
    pos = np.ascontiguousarray(positions, dtype=np.uint32)
    a0 = np.ascontiguousarray(allele0, dtype=np.uint8)
    a1 = np.ascontiguousarray(allele1, dtype=np.uint8)
    q = np.ascontiguousarray(quality, dtype=np.uint32)
    cdef size_t n_ind = a0.shape[0] if a0.ndim == 2 else 0
    cdef size_t n = pos.shape[0]
    if n_ind and (a0.shape[1] != n or a1.shape != a0.shape or q.shape != a0.shape or len(sample_ids) != n_ind):
        raise ValueError("emit_superreads: array shapes disagree")
    # one padded row so that &v[i, 0] is valid for an empty table
    if n == 0:
        pos = np.zeros(1, dtype=np.uint32)
        a0 = np.zeros((n_ind, 1), dtype=np.uint8); a1 = a0; q = np.zeros((n_ind, 1), dtype=np.uint32)
    cdef const uint32_t[::1] v_pos = pos
    cdef const uint8_t[:, ::1] v_a0 = a0
    cdef const uint8_t[:, ::1] v_a1 = a1
    cdef const uint32_t[:, ::1] v_q = q
    sid = np.ascontiguousarray(np.asarray(sample_ids, dtype=np.int64).astype(np.int32))
    cdef const int32_t[::1] v_sid = sid
    cdef size_t i
    cdef ReadSet rs
    cdef vector[cpp.ReadSet*] built
    if n_ind == 0:
        return []
    built.resize(n_ind)
    whamd_emit_superread_sets(n_ind, 1 if numbered else 0, &v_sid[0], n, &v_pos[0], &v_a0[0, 0], &v_a1[0, 0], &v_q[0, 0], built.data())
    results = []
    for i in range(n_ind):
        rs = ReadSet()
        del rs.thisptr
        rs.thisptr = built[i]
        results.append(rs)
    return results
