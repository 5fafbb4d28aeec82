"""Worker of tests/test_blocks_distributed.py: one process per rank (gloo on CPU here, nccl on GPUs).
Every rank derives the same LPT assignment, solves its own blocks (oracle on CPU ranks, HIP path when
WHAMD_TEST_DEVICE=1), results are gathered and rank 0 checks the concatenation against the whole-instance oracle."""
import os
import random
import sys

import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import oracle  # noqa: E402
from helpers import first_difference, table_solution  # noqa: E402
from whatshap_amd import _native  # noqa: E402
from whatshap_amd.blocks import assign_blocks, merge_block_solutions, split_independent_blocks  # noqa: E402
from whatshap_amd.synthetic import random_small_instance  # noqa: E402


def multi_block_instance(seed):
    """Several random single-individual instances laid side by side (disjoint position ranges) as ONE ReadSet."""
    import numpy as np
    rng = random.Random(seed)
    parts = [random_small_instance(rng, mode="single", allow_conflict=False) for _ in range(rng.randint(3, 6))]
    read_ptr, pos, alle, qual, positions, geno, gl, recomb = [0], [], [], [], [], [], [], []
    offset = 0
    for p in parts:
        pos += [int(x) + offset for x in p.var_position]
        alle += p.var_allele.tolist()
        qual += p.var_quality.tolist()
        base = read_ptr[-1]
        read_ptr += [base + int(x) for x in p.read_ptr[1:]]
        positions += [int(x) + offset for x in p.positions]
        geno.append(p.genotype.reshape(1, -1))
        gl.append(p.genotype_likelihoods.reshape(1, -1, 3))
        recomb += p.recombcost.tolist()
        offset = positions[-1] + 100
    distrust = rng.random() < 0.5
    n_reads = len(read_ptr) - 1
    return _native.ProblemArrays(read_ptr, pos, alle, qual, [0] * n_reads, [0], [], np.concatenate(geno, axis=1),
                                 np.concatenate(gl, axis=1), recomb, positions, distrust)


def main():
    dist.init_process_group(backend=os.environ.get("WHAMD_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    use_device = os.environ.get("WHAMD_TEST_DEVICE") == "1"
    failures = []
    for seed in range(12):
        whole = multi_block_instance(seed)
        blocks = split_independent_blocks(whole)
        weights = [float(sum(2 ** 1 for _ in range(b[2][1] - b[2][0]))) * (1 + len(b[1])) for b in blocks]
        mine = assign_blocks(weights, world)[rank]
        local = {}
        for b in mine:
            sub = blocks[b][0]
            # one process per GPU; with fewer devices than ranks (a 1-GPU box) the ranks share device 0
            table = _native.NativeTable(sub, device=rank % max(_native.device_count(), 1)) if use_device else oracle.OracleTable(sub)
            local[b] = table_solution(table)
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        if rank == 0:
            solutions = {}
            for part in gathered:
                assert not (set(part) & set(solutions)), "a block was solved twice"
                solutions.update(part)
            assert sorted(solutions) == list(range(len(blocks))), "a block was not solved"
            merged = merge_block_solutions(whole.n_reads, whole.n_individuals, blocks, solutions)
            want = table_solution(oracle.OracleTable(whole))
            want.pop("sample_ids"), merged.pop("sample_ids")
            if merged != want:
                failures.append((seed, first_difference(want, merged)))
    if rank == 0:
        print("BLOCKS_OK" if not failures else f"BLOCKS_FAILED {failures}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
