"""Test helpers: text matrices -> ReadSets (the input format of the reference's own unit tests,
whatshap/testhelpers.py:18-82), solution tuples of the HIP path / oracle, golden-file I/O."""

import json
import os
import textwrap

import numpy as np

from whatshap_amd import _native
from whatshap_amd.core import Genotype, Read, ReadSet

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def string_to_readset(s, w=None, sample_ids=None, source_id=0, scale_quality=None):
    """One read per line, one character per variant column (' ' = not covered); position = 10*(col+1)."""
    lines = textwrap.dedent(s).strip().split("\n")
    weights = textwrap.dedent(w).strip().split("\n") if w is not None else None
    rs = ReadSet()
    for index, line in enumerate(lines):
        if not line:
            continue
        sample = 0 if sample_ids is None else sample_ids[index]
        read = Read(f"Read {index + 1}", 50, source_id, sample)
        for col, ch in enumerate(line):
            if ch == " ":
                continue
            q = int(weights[index][col]) if weights is not None else 1
            if scale_quality is not None:
                q *= scale_quality
            read.add_variant(position=(col + 1) * 10, allele=int(ch), quality=q)
        assert len(read) > 1, "Reads covering less than two variants are not allowed"
        rs.add(read)
    return rs


def string_to_readset_pedigree(s, w=None, scaling_quality=None):
    """Like string_to_readset, the first character of each line names the individual (A, B, C, ...)."""
    sources, body = [], []
    for line in textwrap.dedent(s).strip().split("\n"):
        if not line:
            continue
        sources.append(ord(line[0]) - ord("A"))
        body.append(line[1:])
    if not body:
        return ReadSet()
    return string_to_readset("\n".join(body), w, sample_ids=sources, scale_quality=scaling_quality)


def matrix_to_readset(lines):
    """'<index> <offset> <alleles> [<offset> <alleles> ...]' per read; position = 10*(offset+i), quality 1."""
    rs = ReadSet()
    for expected_index, line in enumerate(lines, start=1):
        fields = line.split()
        assert len(fields) % 2 == 1 and int(fields[0]) == expected_index, "Not in matrix format."
        read = Read(f"Read {expected_index}", 50)
        for i in range(len(fields) // 2):
            offset = int(fields[2 * i + 1])
            for j, ch in enumerate(fields[2 * i + 2]):
                read.add_variant(position=(offset + j) * 10, allele=int(ch), quality=1)
        rs.add(read)
    return rs


def biallelic_gt(num_alt, ploidy=2):
    return Genotype([0] * (ploidy - num_alt) + [1] * num_alt) if 0 <= num_alt <= ploidy else Genotype([])


def biallelic_gt_list(indices, ploidy=2):
    return [biallelic_gt(i, ploidy) for i in indices]


def table_solution(table):
    """Comparable dict from anything with the NativeTable/OracleTable getters."""
    a0, a1, q, tv, sid = table.super_reads()
    idx, tv2 = table.index_path()
    return {
        "cost": int(table.optimal_score()),
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


def native_solution(problem, path=None, device=0):
    return table_solution(_native.NativeTable(problem, device=device, path=path))


def problem_to_json(p):
    return {
        "read_ptr": p.read_ptr.tolist(),
        "var_position": p.var_position.tolist(),
        "var_allele": p.var_allele.tolist(),
        "var_quality": p.var_quality.tolist(),
        "read_sample_id": p.read_sample_id.tolist(),
        "individual_id": p.individual_id.tolist(),
        "triple_ids": p.triple_ids.tolist(),
        "n_variants": p.n_variants,
        "genotype": p.genotype.tolist(),
        "genotype_likelihoods": None if p.genotype_likelihoods is None else p.genotype_likelihoods.tolist(),
        "recombcost": p.recombcost.tolist(),
        "positions": None if p.positions is None else p.positions.tolist(),
        "distrust_genotypes": p.distrust_genotypes,
    }


def problem_from_json(d):
    gl = d["genotype_likelihoods"]
    return _native.ProblemArrays(
        d["read_ptr"], d["var_position"], d["var_allele"], d["var_quality"], d["read_sample_id"], d["individual_id"],
        d["triple_ids"], np.asarray(d["genotype"], dtype=np.uint8), None if gl is None else np.asarray(gl, dtype=np.float64),
        d["recombcost"], d["positions"], d["distrust_genotypes"], n_variants=d["n_variants"],
    )


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)


def first_difference(a, b):
    for key in a:
        if a[key] != b.get(key):
            return f"{key}: {str(a[key])[:160]} != {str(b.get(key))[:160]}"
    return None


def wmec_cost_of_partitioning(problem, partitioning):
    """Independent numpy evaluation of the single-individual, all-heterozygous wMEC objective for a given read
    bipartition: sum over positions of min(S, W - S), S = weight of the entries that disagree with
    'side 0 carries allele 0, side 1 carries allele 1'.  Size-independent check of a reported optimum."""
    assert problem.n_individuals == 1 and not problem.distrust_genotypes
    lengths = np.diff(problem.read_ptr.astype(np.int64))
    side = np.repeat(np.asarray(partitioning, dtype=np.int64), lengths)
    pos = problem.var_position.astype(np.int64)
    q = problem.var_quality.astype(np.int64)
    disagree = (problem.var_allele.astype(np.int64) != side)
    uniq, inv = np.unique(pos, return_inverse=True)
    s = np.bincount(inv, weights=q * disagree, minlength=uniq.size)
    w = np.bincount(inv, weights=q, minlength=uniq.size)
    return int(np.minimum(s, w - s).sum())
