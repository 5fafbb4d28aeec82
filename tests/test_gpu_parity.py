"""GPU (-m gpu): the HIP path, called through the C ABI, against (i) the reference's known answers through the
drop-in class API, (ii) the committed golden vectors of the compiled reference, (iii) the oracle on seeded inputs,
(iv) size-independent properties at BASELINE.json's full sizes.  Bit-exact everywhere (integer path)."""
import random

import numpy as np
import pytest

import oracle
from helpers import (first_difference, load_golden, native_solution, problem_from_json, table_solution,
                     wmec_cost_of_partitioning)
from reference_cases import all_cases
from whatshap_amd import _native
from whatshap_amd.core import PedigreeDPTable
from whatshap_amd.synthetic import random_small_instance, synthetic_block

pytestmark = pytest.mark.gpu
PATHS = ["auto", "column", "column_keys"]  # auto = resident runs where they apply


def oracle_outcome(problem):
    try:
        return table_solution(oracle.OracleTable(problem)), None
    except oracle.OracleError as e:
        return None, str(e)


def native_outcome(problem, path):
    try:
        return native_solution(problem, path), None
    except _native.SolverError as e:
        return None, str(e)


@pytest.mark.parametrize("case", all_cases(), ids=lambda c: c.name)
def test_reference_unit_tests_through_the_class_api(case):
    """The reference's own assertions (tests/test_phasing.py, tests/test_pedigreephasing.py) on our PedigreeDPTable."""
    dp_table = PedigreeDPTable(case.readset, case.recombcost, case.pedigree, case.distrust_genotypes, case.positions)
    superreads_list, transmission_vector = dp_table.get_super_reads()
    cost = dp_table.get_optimal_cost()
    partition = dp_table.get_optimal_partitioning()
    assert len(superreads_list) == len(case.pedigree)
    assert len(partition) == len(case.readset)
    if case.expected_cost is not None:
        assert cost == case.expected_cost
    if case.constant_transmission:
        assert len(set(transmission_vector)) <= 1
    if case.allowed_transmission is not None:
        assert transmission_vector in case.allowed_transmission
    for superreads in superreads_list:
        assert len(superreads) == 2
        assert [v.position for v in superreads[0]] == [v.position for v in superreads[1]]
    if case.expected_haplotypes is not None:
        for superreads, expected in zip(superreads_list, case.expected_haplotypes):
            haplotypes = tuple(sorted("".join(str(v.allele) for v in sr) for sr in superreads))
            assert haplotypes == tuple(sorted(expected))
    if len(case.pedigree) == 3 and case.expected_haplotypes is not None and case.name.startswith("trio"):
        father, mother, child = superreads_list  # assert_trio_allele_order, tests/test_pedigreephasing.py:46-71
        for pos, tv in enumerate(transmission_vector):
            assert father[int(not (tv % 2))][pos].allele == child[0][pos].allele
            assert mother[int(not (tv // 2))][pos].allele == child[1][pos].allele


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("fixture", ["reference_cases.json", "random_tie_heavy.json", "synthetic_small.json"])
def test_golden_vectors(fixture, path):
    for rec in load_golden(fixture):
        got, err = native_outcome(problem_from_json(rec["problem"]), path)
        assert err == rec["error"], rec["name"]
        if err is None:
            assert got == rec["solution"], f"{rec['name']}: {first_difference(rec['solution'], got)}"


@pytest.mark.parametrize("path", PATHS)
def test_random_tie_heavy_vs_oracle(path):
    rng = random.Random(987 if path == "auto" else 654)
    conflicts = 0
    for i in range(500):
        p = random_small_instance(rng)
        want, werr = oracle_outcome(p)
        got, gerr = native_outcome(p, path)
        assert gerr == werr, i
        conflicts += werr is not None
        assert got == want, f"{i}: {first_difference(want, got) if want else ''}"
    assert conflicts > 0


SYNTHETIC = [
    dict(n_variants=1500, coverage=10, seed=101),
    dict(n_variants=800, coverage=12, seed=102, distrust_genotypes=True),
    dict(n_variants=600, coverage=14, seed=103),
    dict(n_variants=400, coverage=13, seed=104, step=1),
    dict(n_variants=500, coverage=9, seed=105, trio=True),
    dict(n_variants=300, coverage=12, seed=106, trio=True),
    dict(n_variants=200, coverage=9, seed=107, trio=True, distrust_genotypes=True),
    dict(n_variants=300, coverage=8, seed=108, drop_rate=0.6),
    dict(n_variants=300, coverage=3, seed=109),  # every column below one wavefront of projection entries
    dict(n_variants=100000, coverage=15, seed=2, n_columns_limit=120),  # prefix of BASELINE config 2
    dict(n_variants=100000, coverage=15, seed=4, trio=True, n_columns_limit=60),  # prefix of config 4
]


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("kw", SYNTHETIC, ids=str)
def test_synthetic_vs_oracle(kw, path):
    p = synthetic_block(**kw)
    want = table_solution(oracle.OracleTable(p))
    got = native_solution(p, path)
    assert got == want, first_difference(want, got)


def test_coverage_20_prefix_vs_oracle():
    """Prefix of BASELINE config 3 (2^20 bipartitions per column): ~45 columns is ~5 s of oracle time."""
    p = synthetic_block(n_variants=200000, coverage=20, seed=3, n_columns_limit=45)
    want = table_solution(oracle.OracleTable(p))
    got = native_solution(p, "auto")
    assert got == want, first_difference(want, got)


@pytest.mark.parametrize("kw", [dict(n_variants=200000, coverage=22, seed=31, n_columns_limit=58),
                                dict(n_variants=2000, coverage=16, seed=32, distrust_genotypes=True, n_columns_limit=400),
                                dict(n_variants=2000, coverage=17, seed=33, step=3, n_columns_limit=300)], ids=str)
def test_high_coverage_prefixes_vs_oracle(kw):
    """Coverage 22 (10 grid reads + 12-13 local bits: 1024 workgroups per run), three-term costs at coverage 16
    (distrusted genotypes) and a step-3 read layout (two columns without a new read between starts)."""
    p = synthetic_block(**kw)
    want = table_solution(oracle.OracleTable(p))
    for path in ("auto", "column"):
        got = native_solution(p, path)
        assert got == want, (path, first_difference(want, got))


def test_many_reads_ending_at_once():
    """All reads start and end together: k - f jumps from 0 to 12 in one column (the key path's chunked enumeration)."""
    rng = np.random.default_rng(5)
    n_reads, n_var = 12, 6
    read_ptr = np.arange(0, (n_reads + 1) * n_var, n_var)
    pos = np.tile(10 * (np.arange(n_var) + 1), n_reads)
    p = _native.ProblemArrays(read_ptr, pos, rng.integers(0, 2, n_reads * n_var), rng.integers(1, 4, n_reads * n_var),
                              np.zeros(n_reads), [0], [], np.ones((1, n_var)), None, [1] * n_var, None, False)
    want = table_solution(oracle.OracleTable(p))
    for path in PATHS:
        assert native_solution(p, path) == want


@pytest.mark.parametrize("seed", range(4))
def test_four_to_seven_reads_ending_in_one_column_of_a_run(seed):
    """Inside a slot run (not as a per-column step): the third and later ending reads of a column come out of the row's second line."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_slot_plan import _reads_problem

    rng = np.random.default_rng(40 + seed)
    n = 40
    reads = []
    for stop in (9, 17, 26, 33):
        for q in range(4 + seed):
            reads.append((int(stop - 2 - rng.integers(0, 5)), stop))
    for first in range(0, n - 6, 3):
        reads.append((first, min(n - 1, first + int(rng.integers(5, 12)))))
    p = _reads_problem(reads, n, seed)
    want = table_solution(oracle.OracleTable(p))
    summary = _native.plan_summary(p)
    assert summary["n_resident_columns"] >= n - 2, summary
    for path in PATHS:
        assert native_solution(p, path) == want, (path, first_difference(want, native_solution(p, path)))


@pytest.mark.parametrize("kw", [dict(n_variants=50000, coverage=15, seed=2), dict(n_variants=200000, coverage=20, seed=3)], ids=str)
def test_full_size_properties_single_individual(kw):
    """BASELINE configs 2 and 3 at full size (50 000 x coverage 15; 200 000 x coverage 20): the reported optimum
    equals the wMEC objective re-evaluated independently from the reported bipartition; the path is consistent
    between adjacent columns; a second solve reproduces it bit for bit."""
    p = synthetic_block(**kw)
    t = _native.NativeTable(p)
    cost, part = t.optimal_score(), t.partitioning()
    assert wmec_cost_of_partitioning(p, part) == cost
    idx1, tv1 = t.index_path()
    t.solve()
    idx2, tv2 = t.index_path()
    assert t.optimal_score() == cost and (idx1 == idx2).all() and (tv1 == tv2).all()
    a0, a1, q, tv, sid = t.super_reads()
    assert set(np.unique(a0)) <= {0, 1, 3} and ((a0 != a1) | (a0 == 3)).all()  # heterozygous everywhere
    # flipping any single read to the other side must not improve the objective (local optimality of a global optimum)
    rng = np.random.default_rng(1)
    for r in rng.integers(0, p.n_reads, 25):
        flipped = part.copy()
        flipped[r] ^= 1
        assert wmec_cost_of_partitioning(p, flipped) >= cost


def test_full_size_trio_paths_agree():
    """BASELINE config 4 at full size (trio PedMEC, 100 000 SNVs, coverage 15): the fused path and the independent key
    path produce identical cost, backtrace, transmission vector and superreads."""
    p = synthetic_block(n_variants=100000, coverage=15, seed=4, trio=True)
    a = native_solution(p, "column")
    b = native_solution(p, "column_keys")
    assert a == b, first_difference(a, b)
    assert len(set(a["transmission"])) >= 1
    r = native_solution(p, "resident")  # trio runs: LDS-resident slices of T-vectors, per-entry u32 argmin records
    assert r == a, first_difference(a, r)


def test_config5_shape_blocks_on_one_rank():
    """BASELINE config 5 (24 blocks x 100 000 SNVs, coverage 20, over 8 GPUs): one rank's LPT share (3 blocks), each
    solved independently; the optimum of every block equals the independently re-evaluated objective and differs
    between blocks (different seeds); the assignment covers all 24 blocks exactly once."""
    from whatshap_amd.blocks import assign_blocks, block_weight

    shares = assign_blocks([block_weight(100000, 20)] * 24, 8)
    assert sorted(b for r in shares for b in r) == list(range(24)) and all(len(r) == 3 for r in shares)
    costs = []
    for b in shares[0]:
        p = synthetic_block(n_variants=100000, coverage=20, seed=100 + b)
        t = _native.NativeTable(p)
        assert wmec_cost_of_partitioning(p, t.partitioning()) == t.optimal_score()
        costs.append(t.optimal_score())
    assert len(set(costs)) == 3


def test_full_size_resident_equals_column_path():
    """BASELINE config 2 at full size: the resident path (LDS-resident runs, per-workgroup backtrace records) and the
    per-column path (HBM projection column, bit-plane backtrace) agree on every output."""
    p = synthetic_block(n_variants=50000, coverage=15, seed=2)
    a = native_solution(p, "resident")
    b = native_solution(p, "column")
    assert a == b, first_difference(a, b)


@pytest.mark.parametrize("l_pref", [4, 7, 9, 13])
def test_resident_slice_sizes(l_pref):
    """Different grid/local splits of the resident path (1 .. 256 workgroups per run) give identical results."""
    p = synthetic_block(n_variants=700, coverage=14, seed=77)
    want = table_solution(oracle.OracleTable(p))
    t = _native.NativeTable(p, solve=False, path="resident")
    t.set_option("resident_l", str(l_pref))
    t.solve()
    assert table_solution(t) == want


def test_resident_irregular_reads_vs_oracle():
    """Reads of very different lengths, nested and paired-end-like gapped reads: grid reads are not simply the youngest
    reads, several reads end in one column, runs are short and interleave with per-column steps."""
    rng = np.random.default_rng(123)
    n_var = 400
    reads = []
    for start in range(0, n_var - 2):
        for _ in range(rng.integers(0, 3)):
            length = int(rng.choice([2, 3, 5, 9, 17, 30]))
            end = min(n_var, start + length)
            cols = [c for c in range(start, end) if c in (start, end - 1) or rng.random() < 0.7]
            if len(cols) >= 2:
                reads.append(cols)
    read_ptr, pos, alle, qual = [0], [], [], []
    hap = rng.integers(0, 2, n_var)
    for cols in reads:
        side = rng.integers(0, 2)
        for c in cols:
            pos.append(10 * (c + 1))
            alle.append(int(hap[c] ^ side ^ (rng.random() < 0.05)))
            qual.append(int(rng.integers(1, 30)))
        read_ptr.append(len(pos))
    p = _native.ProblemArrays(read_ptr, pos, alle, qual, np.zeros(len(reads)), [0], [], np.ones((1, n_var)), None,
                              [1] * n_var, [10 * (c + 1) for c in range(n_var)], False)
    cov = np.zeros(n_var, dtype=int)
    for cols in reads:
        cov[cols[0]:cols[-1] + 1] += 1
    assert cov.max() <= 25
    want = table_solution(oracle.OracleTable(p))
    for path in PATHS:
        got = native_solution(p, path)
        assert got == want, (path, first_difference(want, got))


def test_blocks_in_flight_concurrently():
    """The host-side work queue: several independent blocks submitted to their own streams on one device before any
    is collected give the same results as solving them one after the other."""
    problems = [synthetic_block(n_variants=6000, coverage=c, seed=200 + i) for i, c in enumerate((15, 12, 18, 15))]
    tables = [_native.NativeTable(p, solve=False) for p in problems]
    for t in tables:
        t.enqueue()
    for t in tables:
        t.wait()
    for p, t in zip(problems, tables):
        want = native_solution(p)
        assert table_solution(t) == want
    # the same through whamd_dptable_enqueue_many (launch sequences interleaved), twice on the same tables, including a
    # trio and an empty table in the batch
    extra = [synthetic_block(n_variants=1500, coverage=10, seed=300, trio=True),
             _native.ProblemArrays([0], [], [], [], [], [0], [], np.zeros((1, 0)), None, [], [], False, n_variants=0)]
    problems += extra
    tables += [_native.NativeTable(p, solve=False) for p in extra]
    for _ in range(2):
        _native.enqueue_many(tables)
        for t in tables:
            t.wait()
        for p, t in zip(problems, tables):
            assert table_solution(t) == native_solution(p)
    with pytest.raises(_native.SolverError):
        _native.enqueue_many([tables[0]])
        tables[0].enqueue()  # already in flight
    tables[0].wait()


def _irregular_problem(seed, n_var, trio, max_cov):
    """Reads of very different lengths, nested and gapped, random samples (trio) -- exotic grid/local layouts."""
    rng = np.random.default_rng(seed)
    cov = np.zeros(n_var, dtype=int)
    read_ptr, pos, alle, qual, samples = [0], [], [], [], []
    hap = rng.integers(0, 2, n_var)
    for start in range(0, n_var - 2):
        for _ in range(int(rng.integers(0, 3))):
            end = min(n_var, start + int(rng.choice([2, 3, 5, 9, 17, 30])))
            if cov[start:end].max() >= max_cov:
                continue
            cols = [c for c in range(start, end) if c in (start, end - 1) or rng.random() < 0.7]
            cov[start:end] += 1
            side = rng.integers(0, 2)
            for c in cols:
                pos.append(10 * (c + 1))
                alle.append(int(hap[c] ^ side ^ (rng.random() < 0.05)))
                qual.append(int(rng.integers(1, 30)))
            read_ptr.append(len(pos))
            samples.append(int(rng.integers(0, 3)) if trio else 0)
    n_ind = 3 if trio else 1
    recomb = [0] + [int(rng.integers(1, 40)) for _ in range(n_var - 1)]
    return _native.ProblemArrays(read_ptr, pos, alle, qual, samples, list(range(n_ind)), [0, 1, 2] if trio else [],
                                 np.ones((n_ind, n_var)), None, recomb, [10 * (c + 1) for c in range(n_var)], False)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_irregular_trio_reads_vs_oracle(seed):
    """Trio runs on reads of irregular lengths: runs are short, several reads end in one column, grid reads are
    scattered over the index, recombination costs vary per column; both device paths must equal the oracle."""
    p = _irregular_problem(seed, 260, True, 12)
    want = table_solution(oracle.OracleTable(p))
    for path in ("auto", "resident", "column"):   # pedigree slot runs, LDS-resident trio runs, per-column kernels
        got = native_solution(p, path)
        assert got == want, (path, first_difference(want, got))
    s = _native.plan_summary(p)
    assert s["n_runs"] > 0 and s["invariants_ok"] == 1


def test_quartet_and_unrelated_individuals_vs_oracle():
    """Two trios sharing parents (T = 16) and a table of unrelated individuals (T = 1, several samples): the
    per-column kernels with their LDS-staged lookup tables."""
    rng = random.Random(77)
    for _ in range(60):
        p = random_small_instance(rng, mode="quartet", max_variants=9, max_reads=7)
        want, werr = oracle_outcome(p)
        got, gerr = native_outcome(p, "auto")
        assert gerr == werr
        assert got == want
    q = synthetic_block(n_variants=300, coverage=9, seed=11, trio=True)
    unrelated = _native.ProblemArrays(q.read_ptr, q.var_position, q.var_allele, q.var_quality, q.read_sample_id, [0, 1, 2],
                                      [], np.ones((3, 300)), None, [1] * 300, q.positions, False)
    want = table_solution(oracle.OracleTable(unrelated))
    assert native_solution(unrelated, "auto") == want


def test_drop_in_class_splits_and_overlaps_independent_blocks():
    """A single-individual table whose reads fall into several disconnected stretches: the drop-in class cuts it at
    the column boundaries no read spans, keeps the blocks in flight on separate streams and concatenates; cost,
    partitioning, superreads and transmission vector equal the whole-table oracle (and the unsplit device solve)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dist_worker import multi_block_instance
    from whatshap_amd.core import Pedigree, NumericSampleIds, Read, ReadSet, Genotype

    for seed in (5, 6, 7, 8):
        whole = multi_block_instance(seed)
        rs = ReadSet()
        for r in range(whole.n_reads):
            read = Read(f"r{r}", 50, 0, 0)
            for i in range(int(whole.read_ptr[r]), int(whole.read_ptr[r + 1])):
                read.add_variant(int(whole.var_position[i]), int(whole.var_allele[i]), int(whole.var_quality[i]))
            rs.add(read)
        ped = Pedigree(NumericSampleIds())
        from whatshap_amd.core import PhredGenotypeLikelihoods
        gl = whole.genotype_likelihoods.reshape(1, -1, 3)
        ped.add_individual(0, [Genotype([0, 1])] * whole.n_variants, [PhredGenotypeLikelihoods(list(gl[0, v])) for v in range(whole.n_variants)])
        ped.numeric_sample_ids.mapping = {0: 0}
        want = table_solution(oracle.OracleTable(whole))
        for split in (True, False):
            dp = PedigreeDPTable(rs, whole.recombcost.tolist(), ped, whole.distrust_genotypes, whole.positions.tolist(), split_blocks=split)
            assert dp.get_optimal_cost() == want["cost"]
            assert dp.get_optimal_partitioning() == want["partitioning"]
            superreads, tv = dp.get_super_reads()
            assert [v.allele for v in superreads[0][0]] == want["allele0"][0]
            assert [v.allele for v in superreads[0][1]] == want["allele1"][0]
            assert [v.quality for v in superreads[0][0]] == want["quality"][0]
            assert [v.position for v in superreads[0][0]] == want["positions"]
            assert tv == want["transmission"]
            assert dp.get_index_path()[0].tolist() == want["index_path"]
        assert dp.get_stats()["n_columns"] == len(want["positions"])


def _reference_cases():
    from reference_cases import all_cases
    return all_cases()


@pytest.mark.parametrize("case", _reference_cases(), ids=[c.name for c in _reference_cases()])
def test_switching_the_reference_class_for_the_drop_in(case):
    """What a WhatsHap user does: the reference's OWN ReadSet / Pedigree objects, the
    reference's PedigreeDPTable on one side, whatshap_amd.shim's replacement (device) on the other; every return value
    of the PhasingAlgorithm interface -- cost, partitioning, superreads as reference ReadSets (names, sample ids,
    positions, alleles, qualities), transmission vector -- is identical."""
    from refobjects import reference_core, table_outputs, to_reference
    from whatshap_amd import shim

    ref = reference_core()
    rs, ped = to_reference(case, ref)
    want = table_outputs(ref.PedigreeDPTable(rs, case.recombcost, ped, case.distrust_genotypes, case.positions))
    table = shim.table_factory(ref)(rs, case.recombcost, ped, case.distrust_genotypes, case.positions)
    superreads, _ = table.get_super_reads()
    assert all(isinstance(x, ref.ReadSet) for x in superreads)
    assert table_outputs(table) == want


@pytest.mark.parametrize("trio", [False, True])
def test_switching_classes_on_synthetic_blocks(trio):
    """Same switch on synthetic blocks large enough for the resident kernels (several runs, folded columns)."""
    from refobjects import problem_to_reference, reference_core, table_outputs
    from whatshap_amd import shim
    from whatshap_amd.synthetic import synthetic_block

    ref = reference_core()
    for seed, coverage, n in ((11, 12, 260), (12, 9, 400)):
        problem = synthetic_block(n, coverage, seed=seed, trio=trio, distrust_genotypes=(seed == 12))
        rs, ped = problem_to_reference(problem, ref)
        recomb = problem.recombcost.tolist()
        positions = None if problem.positions is None else problem.positions.tolist()
        want = table_outputs(ref.PedigreeDPTable(rs, recomb, ped, problem.distrust_genotypes, positions))
        got = table_outputs(shim.table_factory(ref)(rs, recomb, ped, problem.distrust_genotypes, positions))
        assert got == want


def _variant_of(p, quality=None, genotype=None, gl=None, distrust=None, recomb=None):
    """Copy of a ProblemArrays with some inputs replaced."""
    return _native.ProblemArrays(
        p.read_ptr, p.var_position, p.var_allele, p.var_quality if quality is None else quality, p.read_sample_id,
        p.individual_id, p.triple_ids, p.genotype.reshape(p.n_individuals, p.n_variants) if genotype is None else genotype,
        (None if p.genotype_likelihoods is None else p.genotype_likelihoods.reshape(p.n_individuals, p.n_variants, 3)) if gl is None else gl,
        p.recombcost if recomb is None else recomb, p.positions, p.distrust_genotypes if distrust is None else distrust,
        n_variants=p.n_variants)


@pytest.mark.parametrize("seed", [21, 22])
def test_resident_kernel_variants_single_individual(seed):
    """Inputs that steer the single-individual run kernel through each of its evaluation variants, against the oracle:
    heavy weights (Cp + Cm >= 2^14: 32-bit evaluation instead of the packed 16-bit one), homozygous genotypes (a missing
    orientation term), untrusted genotypes with per-column likelihoods (constant term present), two-valued weights
    (ties everywhere, decided by the Gray-rank rule inside the packed path), irregular recombination costs."""
    rng = np.random.default_rng(seed)
    base = synthetic_block(n_variants=420, coverage=13, seed=seed)
    n = base.n_variants
    variants = {
        "heavy": _variant_of(base, quality=base.var_quality * np.uint32(450)),
        "mixed_heavy": _variant_of(base, quality=np.where(rng.random(base.var_quality.size) < 0.3, base.var_quality * np.uint32(900), base.var_quality).astype(np.uint32)),
        "homozygous": _variant_of(base, genotype=rng.choice([0, 1, 1, 2], size=(1, n)).astype(np.uint8)),
        "distrust": _variant_of(base, gl=rng.integers(0, 60, size=(1, n, 3)).astype(np.float64), distrust=True),
        "distrust_heavy": _variant_of(base, quality=base.var_quality * np.uint32(450), gl=rng.integers(0, 9000, size=(1, n, 3)).astype(np.float64), distrust=True),
        "ties": _variant_of(base, quality=(1 + (base.var_quality % 2)).astype(np.uint32)),
    }
    for name, p in variants.items():
        want = table_solution(oracle.OracleTable(p))
        got = native_solution(p, "auto")
        assert got == want, (name, first_difference(want, got))
        assert _native.plan_summary(p)["n_vectorised_columns"] > 300, name


@pytest.mark.parametrize("seed", [31, 32])
def test_resident_kernel_variants_trio(seed):
    """Trio run kernel: genotype combinations with different numbers of cost terms per transmission value, untrusted
    genotypes (16 terms per value: more than the register-resident four, the rest comes from the LDS pool), heavy
    weights, ties, per-column recombination costs."""
    rng = np.random.default_rng(seed)
    base = synthetic_block(n_variants=300, coverage=10, seed=seed, trio=True)
    n = base.n_variants
    # Mendel-consistent genotype triples (father, mother, child)
    triples = np.array([(1, 1, 1), (1, 1, 0), (1, 1, 2), (0, 1, 0), (0, 1, 1), (1, 0, 1), (1, 2, 2), (2, 1, 1), (0, 2, 1), (2, 0, 1), (0, 0, 0), (2, 2, 2)], dtype=np.uint8)
    mixed = triples[rng.integers(0, len(triples), size=n)].T.copy()
    recomb = rng.integers(0, 40, size=n).astype(np.uint32)
    variants = {
        "mixed_genotypes": _variant_of(base, genotype=mixed, recomb=recomb),
        "distrust": _variant_of(base, gl=rng.integers(0, 50, size=(3, n, 3)).astype(np.float64), distrust=True, recomb=recomb),
        "heavy": _variant_of(base, quality=base.var_quality * np.uint32(3000)),
        "ties": _variant_of(base, quality=(1 + (base.var_quality % 2)).astype(np.uint32), recomb=(recomb % 3).astype(np.uint32)),
    }
    for name, p in variants.items():
        want = table_solution(oracle.OracleTable(p))
        for path in ("auto", "resident", "column"):
            got = native_solution(p, path)
            assert got == want, (name, path, first_difference(want, got))
        assert _native.plan_summary(p)["n_resident_columns"] > 200, name
        assert _native.plan_summary(p, "resident")["n_resident_columns"] > 200, name
    # weights so large that the per-individual sums leave the 24-bit range of the term evaluation: the planner must keep
    # those columns away from the run kernel, and the result must still be exact
    small = synthetic_block(n_variants=40, coverage=10, seed=seed, trio=True)
    huge = _variant_of(small, quality=small.var_quality * np.uint32(80000))
    assert _native.plan_summary(huge)["n_resident_columns"] < 40
    assert native_solution(huge, "auto") == table_solution(oracle.OracleTable(huge))


def test_tables_created_and_solved_from_several_host_threads():
    """The library is used from worker threads (one per chromosome / family): creation (flattening, planning, upload) and
    solves running concurrently in four host threads give the single-threaded results; errors stay per thread."""
    import threading

    problems = [synthetic_block(n_variants=3000 + 500 * i, coverage=12 + (i % 3) * 3, seed=400 + i, trio=(i % 4 == 3)) for i in range(8)]
    want = [native_solution(p) for p in problems]
    got = [None] * len(problems)
    errors = []

    def work(indices):
        try:
            for i in indices:
                table = _native.NativeTable(problems[i], solve=False)
                table.solve()
                got[i] = table_solution(table)
                table.close()
            bad = _native.ProblemArrays([0, 2], [20, 10], [0, 1], [1, 1], [0], [0], [], np.ones((1, 2)), None, [1, 1], None, False)
            try:
                _native.NativeTable(bad)
                errors.append("unsorted variants accepted")
            except _native.SolverError as exc:
                if "unsorted" not in str(exc):
                    errors.append(str(exc))
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(list(range(k, len(problems), 4)),)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert got == want


def test_connected_components_as_independent_jobs():
    """A single-individual ReadSet made of many connected components, solved as ONE table: every component but the
    last becomes its own job (forward steps from cost 0, score added on the host, backtrace from entry 0); the jobs are
    spread over lanes that advance in lockstep (batched launches).  Cost, path, partitioning and superreads equal the
    oracle's for the whole ReadSet, for every number of lanes."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from gpu_multiblock import chromosome

    for coverage, n_blocks, seed in ((9, 25, 1), (12, 12, 2)):
        rng = np.random.default_rng(seed)
        whole = chromosome(n_blocks, coverage, seed=seed, max_len=120)
        # irregular weights, some homozygous columns: not every run takes the packed path
        whole = _variant_of(whole, quality=np.where(rng.random(whole.var_quality.size) < 0.2, whole.var_quality * np.uint32(700), whole.var_quality).astype(np.uint32),
                            genotype=rng.choice([0, 1, 1, 1, 2], size=(1, whole.n_variants)).astype(np.uint8))
        want = table_solution(oracle.OracleTable(whole))
        for lanes in (1, 2, 7, 32):
            t = _native.NativeTable(whole, solve=False)
            t.set_option("lanes", str(lanes))
            for _ in range(2):  # solved twice: the lanes' scratch must be re-armed
                t.solve()
                got = table_solution(t)
                assert got == want, (coverage, lanes, first_difference(want, got))
            t.close()
    # through the interleaved multi-table submission as well
    tables = [_native.NativeTable(chromosome(6, 10, seed=s, max_len=80), solve=False) for s in (3, 4, 5)]
    _native.enqueue_many(tables)
    for t in tables:
        t.wait()
    for s, t in zip((3, 4, 5), tables):
        assert table_solution(t) == table_solution(oracle.OracleTable(chromosome(6, 10, seed=s, max_len=80)))


def test_many_small_component_instances_vs_oracle():
    """Fuzz of the job machinery: ReadSets glued from 3-6 random tie-heavy single-individual instances (components of a
    few columns, per-column kernels and runs mixed inside a job, reads of one variant-gap, extra positions), default
    lanes and one lane, both forward paths."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dist_worker import multi_block_instance

    for seed in range(100, 160):
        whole = multi_block_instance(seed)
        want = table_solution(oracle.OracleTable(whole))
        for path, lanes in (("auto", None), ("auto", "1"), ("column", None)):
            t = _native.NativeTable(whole, solve=False, path=path)
            if lanes:
                t.set_option("lanes", lanes)
            t.solve()
            got = table_solution(t)
            assert got == want, (seed, path, lanes, first_difference(want, got))
            t.close()


@pytest.mark.parametrize("seed", [41, 42, 43])
def test_complement_symmetry_on_every_run(seed):
    """Single individual: D[~x] == D[x].  With symmetry=2 every run that has a grid read launches only the workgroups
    whose top grid-read bit is 0, readers fetch the missing half from the complement index, and the backtrace reads the
    mirror-image decisions when the path runs through the half that was not computed.  Everything must equal the oracle
    (and the run with the symmetry switched off), including ties."""
    rng = np.random.default_rng(seed)
    base = synthetic_block(n_variants=500, coverage=14, seed=seed, step=int(rng.integers(1, 4)))
    n = base.n_variants
    variants = {
        "plain": base,
        "ties": _variant_of(base, quality=(1 + (base.var_quality % 2)).astype(np.uint32)),
        "distrust": _variant_of(base, gl=rng.integers(0, 40, size=(1, n, 3)).astype(np.float64), distrust=True),
        "homozygous": _variant_of(base, genotype=rng.choice([0, 1, 1, 2], size=(1, n)).astype(np.uint8)),
        "heavy": _variant_of(base, quality=base.var_quality * np.uint32(450)),
        "irregular": _irregular_problem(seed, 300, False, 13),
    }
    for name, p in variants.items():
        want = table_solution(oracle.OracleTable(p))
        for level in ("2", "0", "1"):
            t = _native.NativeTable(p, solve=False)
            t.set_option("symmetry", level)
            t.solve()
            got = table_solution(t)
            assert got == want, (name, level, first_difference(want, got))
            t.close()


def test_blank_alleles_inside_reads_vs_oracle():
    """Entry::BLANK as the allele of a read's own variant (legal for the reference, src/pedigreecolumncostcomputer.cpp:
    69-70): skipped, its phred ignored -- on every device path."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle import _with_blank_alleles

    rng = random.Random(314)
    for i in range(80):
        p = _with_blank_alleles(random_small_instance(rng, allow_conflict=False), rng)
        want = table_solution(oracle.OracleTable(p))
        for path in PATHS:
            got = native_solution(p, path)
            assert got == want, (i, path, first_difference(want, got))
    big = _with_blank_alleles(synthetic_block(n_variants=500, coverage=13, seed=77), rng)
    assert native_solution(big, "auto") == table_solution(oracle.OracleTable(big))


def test_work_queue_over_several_device_workers():
    """solve_blocks(devices=[...]): LPT assignment of independent blocks to one worker thread per device entry (two
    workers on device 0 here), results in input order and equal to solving every block on its own; a device that does
    not exist is an error, not a fallback."""
    from whatshap_amd.blocks import solve_blocks

    problems = [synthetic_block(n_variants=2500 + 400 * i, coverage=10 + (i % 4) * 2, seed=500 + i) for i in range(7)]
    want = [native_solution(p) for p in problems]
    for devices in ([0], [0, 0], [0, 0, 0]):
        tables = solve_blocks(problems, devices=devices, max_in_flight=2)
        assert [table_solution(t) for t in tables] == want
    with pytest.raises(RuntimeError, match="visible"):
        solve_blocks(problems[:1], devices=[0, _native.device_count()])


def test_two_ranks_on_the_device_path():
    """tests/dist_worker.py with the HIP path (WHAMD_TEST_DEVICE=1): two processes (gloo rendezvous, both on device 0 of
    a 1-GPU box), every rank solves its LPT share on the device, rank 0 checks the concatenation against the
    whole-instance oracle."""
    import os, socket, subprocess, sys

    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", WHAMD_TEST_BACKEND="gloo", WHAMD_TEST_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "dist_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "BLOCKS_OK" in res.stdout


def test_bench_with_two_ranks_on_one_device_matches_a_single_rank():
    """`bench.py --gpus 2` (both ranks on device 0: --oversubscribe) at reduced size: gloo rendezvous, LPT assignment of the
    blocks, no collective on the data path -- the per-rank cost checksums must add up to a single-rank solve of the same
    blocks, every block on exactly one rank."""
    import json, os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--variants", "4000", "--blocks", "6", "--coverage", "14", "--steps", "1", "--warmup", "1", "--pmc", "off", "--cpu-baseline-columns", "0"]
    lines = []
    for gpus in (1, 2):
        res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--oversubscribe"] + common,
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        lines.append(json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]))
        if gpus == 2:
            assert res.stderr.count("[bench rank") == 2   # every rank reports its device, blocks and wall time
    one, two = lines
    assert one["config"]["blocks"] == two["config"]["blocks"] == 6 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert sorted(sum(two["config"]["block_seeds_per_rank"], [])) == sorted(sum(one["config"]["block_seeds_per_rank"], []))
    assert len(two["config"]["optimal_cost_checksum_per_rank"]) == 2 and all(c > 0 for c in two["config"]["optimal_cost_checksum_per_rank"])
    assert sum(two["config"]["optimal_cost_checksum_per_rank"]) == one["config"]["optimal_cost_checksum"] == two["config"]["optimal_cost_checksum"]
    assert "gloo" in two["config"]["rendezvous"]
    # `value` times FRESH tables (create inside the clock), the resident figure rides beside it; every rank reports its create / solve walls and its CPU slice
    for line in (one, two):
        assert line["value_resident"]["value"] > 0 and line["value"] > 0   # (no order between them at this size: six 4 000-column tables, one step)
    assert [r["rank"] for r in two["per_rank"]] == [0, 1] and all(r["create_ms"] >= 0 and r["solve_ms"] > 0 and r["cpus"] >= 1 for r in two["per_rank"])
    assert sum(r["tables"] for r in two["per_rank"]) == 6 and "bound to the CPUs" in (two["config"].get("cpu_binding") or "bound to the CPUs")
    # per-rank checksums against single-rank solves of exactly those blocks
    for seeds, checksum in zip(two["config"]["block_seeds_per_rank"], two["config"]["optimal_cost_checksum_per_rank"]):
        assert checksum == sum(_native.NativeTable(synthetic_block(4000, 14, seed=s)).optimal_score() for s in seeds)


def test_shim_with_compiled_ingestion_on_plain_reference_objects():
    """whatshap_amd.shim.table_factory on WhatsHap's OWN objects without the recording subclass: the compiled ingestion
    (whatshap_amd/ingest, thisptr walk) feeds the device table; every return value of the PhasingAlgorithm interface equals
    the reference PedigreeDPTable's on the same objects."""
    from refobjects import reference_core, table_outputs
    from whatshap_amd import ingest, shim
    from whatshap_amd.synthetic import synthetic_block

    ref = reference_core()
    if ingest.load() is None:
        pytest.skip("whatshap_amd/ingest not built")
    for trio, coverage, n in ((False, 12, 400), (True, 9, 300)):
        problem = synthetic_block(n, coverage, seed=77, trio=trio, distrust_genotypes=trio)
        n_ind = problem.n_individuals
        rs = ref.ReadSet()
        ptr = problem.read_ptr
        for r in range(problem.n_reads):
            read = ref.Read(f"read{r}", 60, 0, int(problem.read_sample_id[r]))
            for i in range(int(ptr[r]), int(ptr[r + 1])):
                read.add_variant(int(problem.var_position[i]), int(problem.var_allele[i]), int(problem.var_quality[i]))
            rs.add(read)
        ids = ref.NumericSampleIds()
        for numeric in range(n_ind):
            assert ids[str(numeric)] == numeric
        ped = ref.Pedigree(ids)   # plain: nothing recorded on the Python side
        gl = None if problem.genotype_likelihoods is None else problem.genotype_likelihoods.reshape(n_ind, problem.n_variants, 3)
        for i in range(n_ind):
            ped.add_individual(str(i), [ref.Genotype([0, 1])] * problem.n_variants,
                               None if gl is None else [ref.PhredGenotypeLikelihoods([float(x) for x in gl[i, v]]) for v in range(problem.n_variants)])
        if trio:
            ped.add_relationship("0", "1", "2")
        recomb, positions = problem.recombcost.tolist(), problem.positions.tolist()
        want = table_outputs(ref.PedigreeDPTable(rs, recomb, ped, problem.distrust_genotypes, positions))
        shim.reset_stats()
        table = shim.table_factory(ref)(rs, recomb, ped, problem.distrust_genotypes, positions)
        assert type(table).__name__ == "_TableAdapter"   # the device table, not the fallback
        assert table_outputs(table) == want
        # the superreads were built in C++ and adopted (whamd_ingest.emit_superreads; whatshap/core.pyx:388-400), no Python loop per variant
        assert shim.stats()["compiled_emits"] == 1 and all(type(x) is ref.ReadSet for x in table.get_super_reads()[0])
