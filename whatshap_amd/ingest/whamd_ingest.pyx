# cython: language_level=3
# distutils: language = c++
"""Compiled ingestion of WhatsHap's own objects: ``whatshap.core.ReadSet`` / ``Pedigree`` -> the flat arrays behind the
C-ABI views (``whatshap_amd._native.ProblemArrays``), walking the C++ objects through ``thisptr`` exactly as
``whatshap/readselect.pyx:14-15,244`` does.  Built against the reference's ``core.pxd`` / ``cpp.pxd`` / ``src/*.h`` by
``whatshap_amd/ingest/build.py`` (as any sibling extension of WhatsHap would be); import ``whatshap.core`` first
(``RTLD_GLOBAL``, as ``whatshap/__init__.py`` does) so that the C++ symbols resolve."""
from libc.stdint cimport int32_t, uint8_t, uint32_t, uint64_t

import numpy as np

from whatshap.core cimport Pedigree, ReadSet
from whatshap cimport cpp


cdef extern from "whamd_ingest_helpers.h":
    size_t whamd_readset_variant_count(cpp.ReadSet*) except +
    void whamd_flatten_readset(cpp.ReadSet*, uint64_t*, int32_t*, uint8_t*, uint32_t*, int32_t*) except +
    int whamd_flatten_pedigree(cpp.Pedigree*, uint32_t*, uint32_t*, uint8_t*, double*) except +


def flatten_readset(ReadSet readset):
    """(read_ptr, var_position, var_allele, var_quality, read_sample_id) of a reference ReadSet."""
    cdef size_t n_reads = readset.thisptr.size()
    cdef size_t nnz = whamd_readset_variant_count(readset.thisptr)
    read_ptr = np.zeros(n_reads + 1, dtype=np.uint64)
    position = np.zeros(max(nnz, 1), dtype=np.int32)
    allele = np.zeros(max(nnz, 1), dtype=np.uint8)
    quality = np.zeros(max(nnz, 1), dtype=np.uint32)
    sample = np.zeros(max(n_reads, 1), dtype=np.int32)
    cdef uint64_t[::1] v_ptr = read_ptr
    cdef int32_t[::1] v_pos = position
    cdef uint8_t[::1] v_allele = allele
    cdef uint32_t[::1] v_quality = quality
    cdef int32_t[::1] v_sample = sample
    whamd_flatten_readset(readset.thisptr, &v_ptr[0], &v_pos[0], &v_allele[0], &v_quality[0], &v_sample[0])
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
