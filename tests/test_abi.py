"""CPU: the C-ABI library loads, exports every symbol include/whatshap_amd.h declares, validates inputs
like the reference does (same messages), and fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import string_to_readset
from whatshap_amd import _native
from whatshap_amd.core import NumericSampleIds, Pedigree, PedigreeDPTable, Read, ReadSet, problem_from_objects
from helpers import biallelic_gt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "whatshap_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(whamd_[a-z_]+)\s*\(", text)))


def header_abi_version():
    text = open(os.path.join(ROOT, "include", "whatshap_amd.h")).read()
    return int(re.search(r"#define WHAMD_ABI_VERSION (\d+)", text).group(1))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_native.EXPORTED_SYMBOLS) == names
    assert lib.whamd_abi_version() == _native.ABI_VERSION == header_abi_version()


def het_pedigree(n_positions):
    ped = Pedigree(NumericSampleIds())
    ped.add_individual("individual0", [biallelic_gt(1)] * n_positions)
    return ped


def test_unsorted_readset_message():
    rs = ReadSet()
    for name, start in (("a", 30), ("b", 10)):
        r = Read(name, 50, 0, 0)
        r.add_variant(start, 0, 1)
        r.add_variant(start + 10, 1, 1)
        rs.add(r)
    with pytest.raises(RuntimeError, match="ColumnIterator: reads in ReadSet are not sorted."):
        PedigreeDPTable(rs, [1] * 4, het_pedigree(4))


def test_unsorted_variants_message():
    rs = ReadSet()
    r = Read("a", 50, 0, 0)
    r.add_variant(20, 0, 1)
    r.add_variant(10, 1, 1)
    rs.add(r)
    with pytest.raises(RuntimeError, match="encountered read with unsorted variants"):
        PedigreeDPTable(rs, [1, 1], het_pedigree(2))


def test_unknown_sample_message():
    rs = ReadSet()
    r = Read("a", 50, 0, 7)
    r.add_variant(10, 0, 1)
    r.add_variant(20, 1, 1)
    rs.add(r)
    with pytest.raises(RuntimeError, match="Individual with ID 7 not present in pedigree."):
        PedigreeDPTable(rs, [1, 1], het_pedigree(2))


def test_mendelian_conflict_message():
    rs = string_to_readset("""
      11
      01
    """, sample_ids=[0, 2])
    ped = Pedigree(NumericSampleIds())
    ped.add_individual("f", [biallelic_gt(0), biallelic_gt(0)])
    ped.add_individual("m", [biallelic_gt(0), biallelic_gt(0)])
    ped.add_individual("c", [biallelic_gt(2), biallelic_gt(1)])  # 1/1 child of two 0/0 parents
    ped.add_relationship("f", "m", "c")
    with pytest.raises(RuntimeError, match="Error: Mendelian conflict"):
        PedigreeDPTable(rs, [1, 1], ped)


def test_coverage_limit_is_an_error_not_a_fallback():
    rs = ReadSet()
    for i in range(26):
        r = Read(f"r{i}", 50, 0, 0)
        r.add_variant(10, i & 1, 1)
        r.add_variant(20, 1, 1)
        rs.add(r)
    with pytest.raises(_native.SolverError) as e:
        PedigreeDPTable(rs, [1, 1], het_pedigree(2))
    assert e.value.status == _native.WHAMD_ERR_UNSUPPORTED


def test_overflow_guard():
    rs = ReadSet()
    r = Read("a", 50, 0, 0)
    r.add_variant(10, 0, 2**31)
    r.add_variant(20, 1, 2**31)
    rs.add(r)
    with pytest.raises(_native.SolverError) as e:
        PedigreeDPTable(rs, [1, 1], het_pedigree(2))
    assert e.value.status == _native.WHAMD_ERR_OVERFLOW


@pytest.mark.skipif(_native.device_count() > 0, reason="only meaningful on a machine without a GPU")
def test_no_gpu_fails_loudly():
    rs = string_to_readset("""
      11
      01
    """)
    with pytest.raises(_native.SolverError) as e:
        PedigreeDPTable(rs, [1, 1], het_pedigree(2))
    assert e.value.status == _native.WHAMD_ERR_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_read_sort_hash_is_libstdcxx_string_hash():
    # std::hash<int> is the identity, so xor-ing source ids must commute like this
    h0 = _native.read_sort_hash("Read 1", 0)
    assert _native.read_sort_hash("Read 1", 5) == h0 ^ 5
    assert _native.read_sort_hash("Read 2", 0) != h0


def test_the_product_library_carries_no_debug_code():
    """the plan emulators, the host instantiation of the heuristic, the kernel instantiations with cycle stamps and the timing
    switches that make results invalid live in libwhatshap_amd_debug.so (test infrastructure), not in what the drop-in classes load."""
    import subprocess

    exported = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "whamd_debug" not in exported
    blob = open(_native.LIB_PATH, "rb").read()
    for marker in (b"WHAMD_SLOT_SKIP", b"WHAMD_SLOT_STAMPS", b"WHAMD_DEBUG_STAMPS", b"WHAMD_NO_YFORM", b"WHAMD_GROUP_PARTS", b"WHAMD_EAGER_TERMS", b"WHAMD_GENERIC_FINISH",
                   b"WHAMD_NO_COMPAT_CACHE", b"WHAMD_HOST_SUPERREADS", b"WHAMD_UPLOAD_ON_TABLE_STREAM", b"WHAMD_UPLOAD_STREAMS_HIGH", b"WHAMD_SPLIT_COMPONENTS", b"WHAMD_SKIP_SLAB_COPY", b"WHAMD_TAIL_OWN_STREAM", b"WHAMD_NO_GROUP_BACKTRACE", b"WHAMD_SYNC_UPLOAD", b"WHAMD_DENSE_COLUMN_UPLOAD", b"WHAMD_NO_WIDE_LAYOUT", b"WHAMD_NO_UPLOAD_SLAB"):
        assert marker not in blob, marker
    debug = subprocess.run(["nm", "-D", "--defined-only", _native.DEBUG_LIB_PATH], capture_output=True, text=True, check=True).stdout
    for name in ("whamd_debug_emulate_slot_plan", "whamd_debug_emulate_pedslot_plan", "whamd_debug_pedmec_heuristic_create_host"):
        assert name in debug


def test_only_the_c_abi_is_exported_and_the_host_pool_keeps_blocks_between_tables():
    """csrc/exports.map + csrc/host_memory.cpp: the shared objects export the C ABI and nothing else -- in particular not the allocation functions the
    library replaces FOR ITSELF (a process that loads it keeps its own malloc / operator new) --, a destroyed table's large arrays stay in the host pool,
    the next create of the same shape reuses them (the pool does not grow), and whamd_release_caches() returns them to the system."""
    import subprocess

    from whatshap_amd.synthetic import synthetic_block

    for path in (_native.LIB_PATH, _native.DEBUG_LIB_PATH):
        if not os.path.exists(path):
            continue
        names = [ln.split()[-1] for ln in subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout.splitlines() if ln.strip()]
        assert names and all(n.startswith("whamd_") for n in names), [n for n in names if not n.startswith("whamd_")][:5]
        assert not any(n.startswith("_Zn") or n.startswith("_Zd") for n in names)     # operator new / delete stay local
    p = synthetic_block(6000, 12, seed=5)
    _native.release_caches()
    assert _native.host_pool_idle_bytes() == 0
    first = _native.plan_summary(p)          # host only: flatten + plan, everything freed at the end of the call
    kept = _native.host_pool_idle_bytes()
    assert kept >= 1 << 20                   # the arrays of 64 KB and more went to the pool (entries alone: 6000 x 12 x 12 B)
    for _ in range(3):
        assert _native.plan_summary(p) == first
    assert _native.host_pool_idle_bytes() == kept      # reused, not grown
    _native.release_caches()
    assert _native.host_pool_idle_bytes() == 0
    assert _native.plan_summary(p) == first


def test_concurrent_creates_share_the_worker_pool():
    """csrc/host_parallel.h WorkerPool: plans made by eight Python threads at once -- each splitting its ranges over pool workers -- equal the plan one
    thread makes alone (ranges are handed out by an atomic counter: every range runs exactly once whoever takes it)."""
    from concurrent.futures import ThreadPoolExecutor

    from whatshap_amd.synthetic import irregular_block, synthetic_block

    problems = [synthetic_block(30000, 10 + i % 4, seed=40 + i) for i in range(6)] + [irregular_block(20000, 12, seed=3), synthetic_block(9000, 9, seed=2, trio=True)]
    alone = [_native.plan_summary(p) for p in problems]
    for _ in range(3):
        with ThreadPoolExecutor(max_workers=8) as pool:
            together = list(pool.map(_native.plan_summary, problems))
        assert together == alone


def test_the_library_is_loaded_once_when_many_threads_ask_first():
    """blocks.solve_blocks' create workers may be the first callers of _native.lib(): sixteen of them at once must get ONE CDLL object (two objects made
    enqueue_many refuse their tables as "of the product and of the debug library")."""
    import subprocess, sys
    code = ("import threading\nfrom whatshap_amd import _native\nout = []\nbarrier = threading.Barrier(16)\n"
            "def ask():\n    barrier.wait()\n    out.append(id(_native.lib()))\n"
            "ts = [threading.Thread(target=ask) for _ in range(16)]\n[t.start() for t in ts]\n[t.join() for t in ts]\nprint(len(set(out)))\n")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, check=True).stdout.strip()
    assert got == "1"
