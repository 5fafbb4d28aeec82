"""Host-side mirror of the part of ``whatshap.core`` that the wMEC/PedMEC path uses.

Same names, argument meaning and error behaviour as the reference's Cython classes
(``whatshap/core.pyx``, ``whatshap/core.pyi``), so the parity tests read like the reference's own
tests (``tests/test_phasing.py``, ``tests/test_pedigreephasing.py``):

    Read, ReadSet, Variant, NumericSampleIds, Genotype, PhredGenotypeLikelihoods, Pedigree,
    PedigreeDPTable

``PedigreeDPTable`` is the drop-in: its constructor flattens the ReadSet/Pedigree into the views of
``include/whatshap_amd.h`` and runs the HIP solver through the C ABI; the three getters of the
``PhasingAlgorithm`` interface (``whatshap/types.py:7-15``) return what the reference returns.
The data classes here are plain Python containers (the reference's are C++ objects behind Cython);
``PedigreeDPTable`` also accepts the reference's own ``whatshap.core.ReadSet`` (anything iterable
that yields reads with ``sample_id`` and variants with ``position/allele/quality``).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _native


@dataclass
class Variant:
    """A single variant on a read (whatshap/variant.py)."""

    position: int
    allele: int
    quality: int


class NumericSampleIds:
    """Mapping of sample names (strings) to numeric ids (core.pyx:25-58)."""

    def __init__(self):
        self.mapping: Dict[str, int] = {}
        self.frozen = False

    def __getitem__(self, sample):
        if not self.frozen and sample not in self.mapping:
            self.mapping[sample] = len(self.mapping)
        return self.mapping[sample]

    def __len__(self):
        return len(self.mapping)

    def __str__(self):
        return str(self.mapping)

    def freeze(self):
        self.frozen = True

    def inverse_mapping(self):
        return {numeric_id: name for name, numeric_id in self.mapping.items()}


class Read:
    """core.pyx:61-272 (src/read.h).  Variants are kept in insertion order until sort()."""

    def __init__(
        self,
        name: Optional[str] = None,
        mapq: int = 0,
        source_id: int = 0,
        sample_id: int = 0,
        reference_start: int = -1,
        BX_tag: Optional[str] = None,
        HP_tag: int = -1,
        PS_tag: int = -1,
        chromosome: Optional[str] = None,
        sub_alignment_id: Optional[str] = None,
        is_supplementary: bool = False,
        reference_end: int = -1,
        is_reverse: bool = False,
    ):
        self.name = name if name is not None else ""
        self.mapqs: Tuple[int, ...] = (mapq,)
        self.source_id = source_id
        self.sample_id = sample_id
        self.reference_start = reference_start
        self.reference_end = reference_end
        self.BX_tag = BX_tag or ""
        self.HP_tag = HP_tag
        self.PS_tag = PS_tag
        self.chromosome = chromosome or ""
        self.sub_alignment_id = sub_alignment_id or ""
        self.is_supplementary = is_supplementary
        self.is_reverse = is_reverse
        self._positions: List[int] = []
        self._alleles: List[int] = []
        self._qualities: List[int] = []

    def _copy(self) -> "Read":
        r = Read(self.name, 0, self.source_id, self.sample_id, self.reference_start, self.BX_tag, self.HP_tag,
                 self.PS_tag, self.chromosome, self.sub_alignment_id, self.is_supplementary, self.reference_end,
                 self.is_reverse)
        r.mapqs = tuple(self.mapqs)
        r._positions = list(self._positions)
        r._alleles = list(self._alleles)
        r._qualities = list(self._qualities)
        return r

    def __repr__(self):
        return (f"Read(name={self.name!r}, mapq={self.mapqs}, source_id={self.source_id}, "
                f"sample_id={self.sample_id}, variants={list(self)})")

    def __iter__(self) -> Iterator[Variant]:
        for i in range(len(self)):
            yield self[i]

    def __len__(self) -> int:
        return len(self._positions)

    def __getitem__(self, key) -> Variant:
        if isinstance(key, slice):
            raise NotImplementedError("Read does not support slices")
        n = len(self)
        if not (-n <= key < n):
            raise IndexError(f"Index out of bounds: {key}")
        if key < 0:
            key += n
        return Variant(self._positions[key], self._alleles[key], self._qualities[key])

    def __setitem__(self, index, variant: Variant):
        n = len(self)
        if not (-n <= index < n):
            raise IndexError(f"Index out of bounds: {index}")
        if index < 0:
            index += n
        if not isinstance(variant, Variant):
            raise ValueError(f"Expected instance of Variant, but found {type(variant)}")
        self._positions[index] = variant.position
        self._alleles[index] = variant.allele
        self._qualities[index] = variant.quality

    def __contains__(self, position) -> bool:
        return position in self._positions

    def add_variant(self, position: int, allele: int, quality: int):
        self._positions.append(int(position))
        self._alleles.append(int(allele))
        self._qualities.append(int(quality))

    def add_mapq(self, mapq: int):
        self.mapqs = self.mapqs + (mapq,)

    def sort(self):
        """Read::sortVariants (src/read.cpp:63-72): by position; duplicates are an error."""
        order = sorted(range(len(self)), key=lambda i: self._positions[i])
        self._positions = [self._positions[i] for i in order]
        self._alleles = [self._alleles[i] for i in order]
        self._qualities = [self._qualities[i] for i in order]
        for i in range(1, len(self)):
            if self._positions[i - 1] == self._positions[i]:
                raise RuntimeError(f"Duplicate variant in read {self.name} at position {self._positions[i]}")

    def is_sorted(self) -> bool:
        return all(self._positions[i - 1] < self._positions[i] for i in range(1, len(self)))

    def has_BX_tag(self) -> bool:
        return self.BX_tag != ""


class ReadSet:
    """core.pyx:274-361 (src/readset.h)."""

    def __init__(self):
        self._reads: List[Read] = []
        self._names: Dict[Tuple[int, str], int] = {}

    def add(self, read: Read):
        key = (read.source_id, read.name)
        if key in self._names:
            raise RuntimeError("ReadSet::add: duplicate read name.")
        self._names[key] = len(self._reads)
        self._reads.append(read._copy())  # the reference copies the wrapped C++ Read as well

    def __iter__(self) -> Iterator[Read]:
        return iter(list(self._reads))

    def __len__(self) -> int:
        return len(self._reads)

    def __getitem__(self, key):
        if isinstance(key, slice):
            raise NotImplementedError("ReadSet does not support slices")
        if isinstance(key, int):
            return self._reads[key]
        if isinstance(key, str):
            raise NotImplementedError("Querying a ReadSet by read name is deprecated, please query by (source_id, name) instead")
        if isinstance(key, tuple) and len(key) == 2 and isinstance(key[0], int) and isinstance(key[1], str):
            if key not in self._names:
                raise KeyError(key)
            return self._reads[self._names[key]]
        raise AssertionError(f"Invalid key: {key}")

    def __str__(self):
        return "ReadSet:\n" + "".join(f"  {i:5d} {r!r}\n" for i, r in enumerate(self._reads))

    def sort(self):
        """ReadSet::sort (src/readset.cpp:41-51) with the comparator of src/readset.h:39-66: reads without
        variants first, then by first position, ties by std::hash(name) ^ std::hash(source_id), then by
        name / source_id.  The hash is libstdc++'s, obtained from the native library."""

        def key(r: Read):
            h = _native.read_sort_hash(r.name, r.source_id)
            if len(r) == 0:
                return (0, 0, h, r.name.encode("utf-8"), r.source_id)
            return (1, r._positions[0], h, r.name.encode("utf-8"), r.source_id)

        if all(len(r) == 0 for r in self._reads):
            self._reads.sort(key=lambda r: (_native.read_sort_hash(r.name, r.source_id), r.name.encode("utf-8"), r.source_id))
        else:
            self._reads.sort(key=key)
        self._names = {(r.source_id, r.name): i for i, r in enumerate(self._reads)}

    def subset(self, reads_to_select: Iterable[int]) -> "ReadSet":
        result = ReadSet()
        for i in sorted(set(int(i) for i in reads_to_select)):  # IndexSet is an ordered std::set<int>
            result.add(self._reads[i])
        return result

    def get_positions(self) -> List[int]:
        positions = set()
        for r in self._reads:
            positions.update(r._positions)
        return sorted(positions)


class Genotype:
    """core.pyx Genotype (src/genotype.h): an unordered multiset of alleles."""

    def __init__(self, alleles: Sequence[int]):
        if len(alleles) >= 15:
            raise RuntimeError("Error: Maximum ploidy for genotype exceeded!")
        if any(a >= 16 for a in alleles):
            raise RuntimeError("Error: Maximum alleles for genotype exceeded!")
        self._alleles = tuple(sorted(int(a) for a in alleles))

    def as_vector(self) -> List[int]:
        return list(self._alleles)

    def is_none(self) -> bool:
        return len(self._alleles) == 0

    def get_ploidy(self) -> int:
        return len(self._alleles)

    def is_homozygous(self) -> bool:
        return len(self._alleles) > 0 and len(set(self._alleles)) == 1

    def is_diploid_and_biallelic(self) -> bool:
        return len(self._alleles) == 2 and all(a <= 1 for a in self._alleles)

    def get_index(self) -> int:
        # src/genotype.cpp:82-93; for diploid bi-allelic genotypes this is the number of ALT alleles
        from math import comb

        index = 0
        for k, allele in enumerate(self._alleles, start=1):
            index += comb(k + allele - 1, allele - 1) if allele >= 1 else 0
        return index

    def _code(self) -> int:
        """WHAMD genotype code of include/whatshap_amd.h."""
        return sum(self._alleles) if self.is_diploid_and_biallelic() else _native.GT_OTHER

    def __str__(self):
        return "." if self.is_none() else "/".join(str(a) for a in self._alleles)

    def __repr__(self):
        return str(self)

    def __eq__(self, other):
        return isinstance(other, Genotype) and self._alleles == other._alleles

    def __hash__(self):
        return hash(self._alleles)

    def __lt__(self, other):
        return self.get_index() < other.get_index()


class PhredGenotypeLikelihoods:
    """core.pyx PhredGenotypeLikelihoods (src/phredgenotypelikelihoods.h)."""

    def __init__(self, gl: Sequence[float], ploidy: int = 2, nr_alleles: int = 2):
        from math import comb

        if comb(ploidy + nr_alleles - 1, nr_alleles - 1) != len(gl):
            raise RuntimeError("Error: wrong number of given genotype likelihoods given.")
        self._gl = [float(x) for x in gl]
        self.ploidy = ploidy
        self.nr_alleles = nr_alleles

    def __getitem__(self, genotype: Genotype) -> float:
        assert genotype.is_diploid_and_biallelic()
        return self._gl[genotype.get_index()]

    def __len__(self):
        return len(self._gl)

    def as_vector(self) -> List[float]:
        return list(self._gl)

    def genotypes(self) -> List[Genotype]:
        assert self.ploidy == 2 and self.nr_alleles == 2
        return [Genotype([0, 0]), Genotype([0, 1]), Genotype([1, 1])]

    def __iter__(self):
        for g in self.genotypes():
            yield self[g]

    def __eq__(self, other):
        return isinstance(other, PhredGenotypeLikelihoods) and self._gl == other._gl

    def __str__(self):
        return "PhredGenotypeLikelihoods(" + ",".join(str(x) for x in self._gl)


class Pedigree:
    """core.pyx:419-467 (src/pedigree.h)."""

    def __init__(self, numeric_sample_ids: NumericSampleIds):
        self.numeric_sample_ids = numeric_sample_ids
        self._ids: List[int] = []
        self._genotypes: List[List[Genotype]] = []
        self._gls: List[List[Optional[PhredGenotypeLikelihoods]]] = []
        self._triples: List[Tuple[int, int, int]] = []  # by numeric id, as Pedigree::addRelationship receives them

    def add_individual(self, id, genotypes: Sequence[Genotype], genotype_likelihoods=None):
        genotypes = list(genotypes)
        if genotype_likelihoods:
            gls = list(genotype_likelihoods)
        else:
            gls = [None] * len(genotypes)
        if self._genotypes:
            assert len(genotypes) == len(self._genotypes[0])
        assert len(gls) == len(genotypes)
        self._ids.append(self.numeric_sample_ids[id])
        self._genotypes.append(genotypes)
        self._gls.append(gls)

    def add_relationship(self, father_id, mother_id, child_id):
        ids = tuple(self.numeric_sample_ids[x] for x in (father_id, mother_id, child_id))
        for i in ids:
            if i not in self._ids:
                raise RuntimeError(f"Individual with ID {i} not present in pedigree.")
        self._triples.append(ids)

    @property
    def variant_count(self) -> int:
        return len(self._genotypes[0]) if self._genotypes else -1

    def _index(self, sample_id) -> int:
        numeric = self.numeric_sample_ids[sample_id]
        for i in range(len(self._ids) - 1, -1, -1):
            if self._ids[i] == numeric:
                return i
        raise RuntimeError(f"Individual with ID {numeric} not present in pedigree.")

    def genotype(self, sample_id, variant_index: int) -> Genotype:
        return self._genotypes[self._index(sample_id)][variant_index]

    def genotype_likelihoods(self, sample_id, variant_index: int):
        return self._gls[self._index(sample_id)][variant_index]

    def __len__(self):
        return len(self._ids)

    def __str__(self):
        return f"Pedigree(individuals={self._ids}, triples={self._triples})"


def _flatten_readset(readset) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """ReadSet (ours or the reference's) -> CSR arrays of the readset view."""
    read_ptr = [0]
    pos: List[int] = []
    alle: List[int] = []
    qual: List[int] = []
    samples: List[int] = []
    for read in readset:
        samples.append(read.sample_id)
        if isinstance(read, Read):
            pos.extend(read._positions)
            alle.extend(read._alleles)
            qual.extend(read._qualities)
        else:  # duck-typed (e.g. whatshap.core.Read)
            for v in read:
                pos.append(v.position)
                alle.append(v.allele)
                qual.append(v.quality)
        read_ptr.append(len(pos))
    alle_arr = np.asarray(alle, dtype=np.int64)
    if alle_arr.size and (alle_arr.min() < 0 or alle_arr.max() > 255):
        raise RuntimeError("read allele must be 0 (REF), 1 (ALT) or 2 (BLANK)")
    return (np.asarray(read_ptr, dtype=np.uint64), np.asarray(pos, dtype=np.int32), alle_arr.astype(np.uint8),
            np.asarray(qual, dtype=np.uint32), np.asarray(samples, dtype=np.int32))


def problem_from_objects(readset, recombcost, pedigree: Pedigree, distrust_genotypes: bool = False,
                         positions=None) -> _native.ProblemArrays:
    """The arrays behind the C-ABI views for (ReadSet, recombcost, Pedigree, distrust, positions)."""
    read_ptr, pos, alle, qual, samples = _flatten_readset(readset)
    n_ind = len(pedigree)
    n_var = max(pedigree.variant_count, 0) if n_ind else 0
    genotype = np.zeros((n_ind, n_var), dtype=np.uint8)
    any_gl = any(gl is not None for gls in pedigree._gls for gl in gls)
    gl = np.full((n_ind, n_var, 3), np.nan, dtype=np.float64) if any_gl else None
    for i in range(n_ind):
        for v in range(n_var):
            genotype[i, v] = pedigree._genotypes[i][v]._code()
            g = pedigree._gls[i][v]
            if g is not None:
                if len(g) != 3:
                    raise RuntimeError("only diploid bi-allelic genotype likelihoods are supported")
                gl[i, v, :] = g.as_vector()
    triples = np.asarray(pedigree._triples, dtype=np.uint32).reshape(-1)
    return _native.ProblemArrays(
        read_ptr, pos, alle, qual, samples, np.asarray(pedigree._ids, dtype=np.uint32), triples, genotype, gl,
        np.asarray(list(recombcost), dtype=np.uint32),
        None if positions is None else np.asarray(list(positions), dtype=np.uint32),
        distrust_genotypes, n_variants=n_var,
    )


def problem_from_reference_objects(ingest, readset, recombcost, pedigree, distrust_genotypes: bool = False,
                                   positions=None) -> _native.ProblemArrays:
    """The same arrays from WhatsHap's OWN ``ReadSet`` / ``Pedigree`` objects through the compiled ingestion
    (``whatshap_amd.ingest``): the C++ objects are walked through ``thisptr`` (``whatshap/readselect.pyx:14-15,244`` is the
    precedent), no Python object is created per variant and the pedigree needs no recording subclass."""
    read_ptr, pos, alle, qual, samples = ingest.flatten_readset(readset)
    ids, triples, genotype, gl = ingest.flatten_pedigree(pedigree)
    u32 = getattr(ingest, "u32_array", lambda values: np.asarray(list(values), dtype=np.uint32))
    return _native.ProblemArrays(
        read_ptr, pos, alle, qual, samples, ids, triples, genotype, gl, u32(recombcost),
        None if positions is None else u32(positions), distrust_genotypes,
        n_variants=genotype.shape[1] if genotype.ndim == 2 else 0,
    )


class PedigreeDPTable:
    """Drop-in for ``whatshap.core.PedigreeDPTable`` (core.pyx:364-416) running on an MI355X.

    ``PedigreeDPTable(readset, recombcost, pedigree, distrust_genotypes=False, positions=None)``;
    all the work happens in the constructor, as in the reference (src/pedigreedptable.cpp:36).
    Extra keyword-only arguments: ``device`` (HIP device index), ``path`` (solver variant, see
    ``whamd_dptable_set_option``) and ``split_blocks`` (default off): a table without trios is cut wherever no read
    is active across a column boundary (exact, including tie-breaks -- whatshap_amd/blocks.py) and the independent
    blocks go through the host-side work queue, ``max_in_flight`` at a time on their own streams.  Off by default
    because one table already runs its connected components side by side on the device
    while every extra table costs ~4 ms of allocation and stream set-up (200 components of coverage 15, 160k columns:
    0.012 s as one table, 1.0 s as 200 tables); separate tables pay off for few, large blocks (bench.py --blocks-per-gpu).
    """

    def __init__(self, readset, recombcost, pedigree: Pedigree, distrust_genotypes: bool = False, positions=None,
                 *, device: int = 0, path: Optional[str] = None, split_blocks: bool = False, max_in_flight: int = 8,
                 problem: Optional[_native.ProblemArrays] = None):
        self.pedigree = pedigree
        # `problem`: the flat arrays are already there (compiled ingestion of reference objects, whatshap_amd.shim)
        self._problem = problem if problem is not None else problem_from_objects(readset, recombcost, pedigree, distrust_genotypes, positions)
        self._tables: List[_native.NativeTable] = []
        self._blocks = None
        blocks = None
        if split_blocks and self._problem.triple_ids.size == 0 and self._problem.n_individuals == 1:
            from .blocks import split_independent_blocks

            firsts = self._problem.var_position[self._problem.read_ptr[:-1][np.diff(self._problem.read_ptr) > 0].astype(np.int64)]
            if np.any(np.diff(firsts.astype(np.int64)) < 0):  # the check the whole-table constructor would have made
                raise RuntimeError("ColumnIterator: reads in ReadSet are not sorted.")
            blocks = split_independent_blocks(self._problem)
            if len(blocks) <= 1:
                blocks = None
        if blocks is None:
            self._tables = [_native.NativeTable(self._problem, device=device, path=path, solve=True)]
            return
        self._blocks = blocks
        from .blocks import solve_blocks

        self._tables = solve_blocks([sub for sub, _, _ in blocks], device=device, path=path, max_in_flight=max_in_flight)

    def _merged(self):
        """(allele0, allele1, quality, transmission, sample ids, positions) of the whole table."""
        parts = [t.super_reads() for t in self._tables]
        positions = np.concatenate([t.positions() for t in self._tables]) if self._tables else np.zeros(0, np.uint32)
        a0 = np.concatenate([p[0] for p in parts], axis=1)
        a1 = np.concatenate([p[1] for p in parts], axis=1)
        q = np.concatenate([p[2] for p in parts], axis=1)
        tv = np.concatenate([p[3] for p in parts])
        return a0, a1, q, tv, parts[0][4], positions

    def raw_super_reads(self):
        """The superreads as arrays, for a compiled emitter (``whatshap_amd.ingest.emit_superreads``): ``(positions u32 [n],
        allele0 u8 [individuals, n], allele1, quality u32 [individuals, n], sample ids, transmission vector, numbered names)``."""
        a0, a1, q, tv, sid, positions = self._merged()
        return positions, a0, a1, q, sid, tv, True

    def get_super_reads(self) -> Tuple[List[ReadSet], List[int]]:
        """Optimal-score haplotypes as one ReadSet of two superreads per individual, plus the
        transmission vector (core.pyx:381-404, src/pedigreedptable.cpp:344-388)."""
        a0, a1, q, tv, sid, positions = self._merged()
        results = []
        position_list = positions.tolist()
        for i in range(len(sid)):
            rs = ReadSet()
            quality_list = q[i].tolist()
            for h, alleles in ((0, a0), (1, a1)):
                read = Read(f"superread_{h}_{i}", -1, -1, int(sid[i]))
                read._positions = position_list  # ReadSet.add copies the read (and its lists)
                read._alleles = alleles[i].tolist()
                read._qualities = quality_list
                rs.add(read)
            results.append(rs)
        return results, tv.tolist()

    def get_optimal_cost(self) -> int:
        return sum(t.optimal_score() for t in self._tables)

    def get_optimal_partitioning(self) -> List[int]:
        if self._blocks is None:
            return self._tables[0].partitioning().tolist()
        out = [1] * self._problem.n_reads
        for (_, reads, _), t in zip(self._blocks, self._tables):
            part = t.partitioning().tolist()  # one native call (and one copy) per block
            for local, r in enumerate(reads):
                out[int(r)] = part[local]
        return out

    # not part of the reference API: raw backtrace and device measurements
    def get_index_path(self):
        idx = np.concatenate([t.index_path()[0] for t in self._tables])
        tv = np.concatenate([t.index_path()[1] for t in self._tables])
        return idx, tv

    def get_stats(self) -> dict:
        stats = [t.stats() for t in self._tables]
        out = dict(stats[0])
        for s in stats[1:]:
            for key in ("n_columns", "n_cells", "n_costs", "algorithmic_bytes", "forward_launches", "forward_ms",
                        "backtrace_ms", "total_ms", "host_prepare_ms", "host_finish_ms"):
                out[key] += s[key]
            out["max_coverage"] = max(out["max_coverage"], s["max_coverage"])
        out["blocks"] = len(stats)
        return out
