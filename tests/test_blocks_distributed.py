"""CPU: the multi-GPU path's host logic -- exact block splitting, LPT assignment, per-rank solve, gather and
concatenation -- with world_size 2 over gloo (the data path itself has no collective)."""
import os
import random
import socket
import subprocess
import sys

import oracle
from helpers import first_difference, table_solution
from whatshap_amd.blocks import assign_blocks, block_weight, merge_block_solutions, split_independent_blocks

HERE = os.path.dirname(os.path.abspath(__file__))


def test_lpt_assignment_is_deterministic_and_balanced():
    weights = [block_weight(100000, 20)] * 24
    a = assign_blocks(weights, 8)
    assert a == assign_blocks(weights, 8)
    assert sorted(b for r in a for b in r) == list(range(24)) and all(len(r) == 3 for r in a)
    uneven = assign_blocks([8, 7, 6, 5, 4], 2)
    loads = [sum([8, 7, 6, 5, 4][b] for b in r) for r in uneven]
    assert abs(loads[0] - loads[1]) <= 4 and uneven[0][0] == 0 and uneven[1][0] == 1


def test_split_and_merge_single_process():
    sys.path.insert(0, HERE)
    from dist_worker import multi_block_instance
    n_multi = 0
    for seed in range(40):
        whole = multi_block_instance(1000 + seed)
        blocks = split_independent_blocks(whole)
        n_multi += len(blocks) > 1
        sols = {b: table_solution(oracle.OracleTable(blk[0])) for b, blk in enumerate(blocks)}
        merged = merge_block_solutions(whole.n_reads, whole.n_individuals, blocks, sols)
        want = table_solution(oracle.OracleTable(whole))
        merged.pop("sample_ids"), want.pop("sample_ids")
        assert merged == want, first_difference(want, merged)
    assert n_multi >= 30


def test_world_size_2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", WHAMD_TEST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dist_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "BLOCKS_OK" in res.stdout
