"""Host-side sharding of independent phasing blocks over the GPUs of a node (no collective on the data path).

Two levels, both exact:

* separate PedigreeDPTable instances (chromosomes, families) are independent by construction
  (whatshap/cli/phase.py:467,486,604);
* *within* a single-individual instance (T = 1) the column chain can be cut wherever no read is active across
  a column boundary (f_c = 0): the projection there collapses to one scalar (src/pedigreedptable.cpp:319-325)
  that is added to every cell of the next column (:274-283), so arg-minima -- including the Gray-code tie-breaks
  -- are unchanged; total cost = sum of block costs, index path / partitioning / superreads = concatenation.
  With trios (T > 1) blocks are coupled through the transmission vector; such instances are never split.

Blocks are assigned longest-processing-time-first to the least loaded rank (the same heuristic the reference
uses for its only worker pool, whatshap/polyphase/algorithm.py:101-128).  Each rank (one process per GPU) solves
its blocks; results are concatenated on the host.
"""

from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from ._native import ProblemArrays


def block_weight(n_columns: int, coverage: int, transmissions: int = 1) -> float:
    """Work estimate of a block: number of bipartition costs."""
    return float(n_columns) * float(2 ** coverage) * float(transmissions) ** 2


def assign_blocks(weights: Sequence[float], world_size: int) -> List[List[int]]:
    """LPT: blocks sorted by descending weight (ties by index) go to the currently least loaded rank
    (ties by rank).  Deterministic, so every rank computes the same assignment without communicating."""
    order = sorted(range(len(weights)), key=lambda b: (-weights[b], b))
    load = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for b in order:
        r = min(range(world_size), key=lambda i: (load[i], i))
        out[r].append(b)
        load[r] += weights[b]
    return out


def _column_spans(problem: ProblemArrays) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    positions = problem.positions
    if positions is None:
        positions = np.unique(problem.var_position.astype(np.int64)).astype(np.uint32)
    ptr = problem.read_ptr.astype(np.int64)
    first = np.searchsorted(positions, problem.var_position[ptr[:-1]].astype(np.int64)) if problem.n_reads else np.zeros(0, np.int64)
    last = np.searchsorted(positions, problem.var_position[ptr[1:] - 1].astype(np.int64)) if problem.n_reads else np.zeros(0, np.int64)
    return positions, first, last


def split_independent_blocks(problem: ProblemArrays) -> List[Tuple[ProblemArrays, np.ndarray, Tuple[int, int]]]:
    """Cuts a single-individual problem at every column boundary no read is active across.
    Returns [(sub-problem, read indices of the block, (first column, end column))]."""
    if problem.triple_ids.size:
        raise ValueError("instances with trios are coupled through the transmission vector and cannot be split")
    positions, first, last = _column_spans(problem)
    n = positions.size
    crossing = np.zeros(n + 1, dtype=np.int64)  # crossing[c] = reads active in both column c-1 and c
    np.add.at(crossing, first + 1, 1)
    np.add.at(crossing, last + 1, -1)
    crossing = np.cumsum(crossing)[:n]
    starts = [0] + [c for c in range(1, n) if crossing[c] == 0]
    bounds = list(zip(starts, starts[1:] + [n]))
    ptr = problem.read_ptr.astype(np.int64)
    geno = problem.genotype.reshape(problem.n_individuals, problem.n_variants)
    gl = None if problem.genotype_likelihoods is None else problem.genotype_likelihoods.reshape(problem.n_individuals, problem.n_variants, 3)
    recomb = np.zeros(n, dtype=np.uint32)
    m = min(n, problem.recombcost.size)
    recomb[:m] = problem.recombcost[:m]
    if m < n:
        recomb[m:] = problem.recombcost[-1] if problem.recombcost.size else 0
    # block of every read = block of its first column; reads of a block keep their ReadSet order
    block_of = np.searchsorted(np.asarray(starts), first, side="right") - 1
    order = np.argsort(block_of, kind="stable")
    cuts = np.searchsorted(block_of[order], np.arange(len(bounds) + 1))
    lengths_all = (ptr[1:] - ptr[:-1]).astype(np.int64)
    out = []
    for b, (c0, c1) in enumerate(bounds):
        reads = order[cuts[b]:cuts[b + 1]]
        lengths = lengths_all[reads]
        if reads.size and np.all(np.diff(reads) == 1):  # the usual case: a contiguous stretch of the sorted ReadSet
            sel = slice(int(ptr[reads[0]]), int(ptr[reads[-1] + 1]))
        else:
            sel = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in reads]) if reads.size else np.zeros(0, np.int64)
        sub_ptr = np.zeros(reads.size + 1, dtype=np.uint64)
        sub_ptr[1:] = np.cumsum(lengths)
        sub = ProblemArrays(sub_ptr, problem.var_position[sel], problem.var_allele[sel], problem.var_quality[sel],
                            problem.read_sample_id[reads], problem.individual_id, problem.triple_ids, geno[:, c0:c1],
                            None if gl is None else gl[:, c0:c1, :], recomb[c0:c1], positions[c0:c1],
                            problem.distrust_genotypes, n_variants=c1 - c0)
        out.append((sub, reads, (c0, c1)))
    return out


def solve_blocks(problems: Sequence[ProblemArrays], device: int = 0, path=None, max_in_flight: int = 8,
                 release: bool = True, devices: Sequence[int] = None, weights: Sequence[float] = None, create_threads: int = 16,
                 windows_on_device: int = 1, trace: list = None, eager_create: bool = False, host_threads_per_create: int = 2):
    """Host-side work queue (BASELINE north_star: "independent phasing blocks shard across the GPUs of one node via a
    host-side work queue"; scheduling precedent: whatshap/polyphase/algorithm.py:101-128).

    One device (``devices`` None): solves independent blocks ``max_in_flight`` at a time through
    ``whamd_dptable_enqueue_many`` -- tables on slot runs share their launches (one launch per super-step serves the whole
    window: a coverage-15 table alone is 8 workgroups on 256 CUs), up to four full-width tables keep their own streams --
    while ``create_threads`` host threads build the tables of the next window.  ``windows_on_device=2`` collects a window only after
    the next one has been submitted and ``eager_create`` queues every create at once; both were measured (scripts/gpu_e2e_trace.py,
    24 coverage-15 tables) and gain nothing on a 32-thread host: a window costs about 35 ms from enqueue to collect whatever its size
    (the length of the launch sequence) and the creates are bound by the host's cores (24 of them: 35 ms), so one window of
    everything is the fastest schedule there.  ``create_threads`` x ``host_threads_per_create`` is the host parallelism of the creates (16 x 2 by
    default; bench.py also tries one worker per table with four threads each and reports what it used).  ``trace``: a list that receives
    (event, window, ms) tuples.

    Several devices (``devices=[0, 1, ...]``; an index may repeat: two workers on one device): the blocks are assigned
    longest-processing-time-first to the least loaded device (``assign_blocks``; ``weights`` defaults to the number of
    variant entries of a block), one worker thread per entry of ``devices`` runs the single-device queue on its share
    (the C library is thread-safe and ctypes releases the GIL inside every call).  No collective: the results are
    concatenated on the host.  Raises if a requested device does not exist -- there is no fallback.

    Returns the solved tables in input order; with ``release`` their device buffers and streams are freed as soon as
    the solution is on the host."""
    from ._native import NativeTable, device_count, enqueue_many, wait_many

    problems = list(problems)
    if devices is None:
        # create (flatten + plan + upload: host work, the C library releases the GIL) of the NEXT window runs on a few threads while the
        # device solves the current one; the windows' host-side result extraction runs in parallel as well (wait_many)
        from concurrent.futures import ThreadPoolExecutor

        windows = [problems[start:start + max_in_flight] for start in range(0, len(problems), max_in_flight)]
        tables = []
        n_workers = max(1, min(create_threads, max_in_flight))
        with ThreadPoolExecutor(max_workers=n_workers) as pool:
            def options_of(window):
                opts = {}
                if len(window) > 4:       # more than four tables per window share their launches: the library picks the layout for that
                    opts["shared_launches"] = "1"
                if n_workers > 1 and len(window) > 1:   # several creates at once: each keeps to a few threads of its own (32 each would fight)
                    opts["host_threads"] = str(max(1, int(host_threads_per_create)))
                return opts or None

            def create(sub, opts):
                return NativeTable(sub, device=device, path=path, solve=False, options=opts)

            def submit_window(window):
                opts = options_of(window)
                return [pool.submit(create, sub, opts) for sub in window]

            all_pending = [submit_window(w) for w in windows] if eager_create else None   # every create queued at once, in table order
            pending = (all_pending[0] if eager_create else submit_window(windows[0])) if windows else []
            releases = []
            in_flight = None              # the window submitted before this one: collected AFTER the next one is on the device

            def collect(window):
                wait_many(window)         # (device + the host-side result extraction, on several threads)
                if release:               # (stream synchronisation, buffers back to the pools: off the critical path)
                    releases.extend(pool.submit(t.release_device) for t in window)
                tables.extend(window)

            import time
            t_begin = time.perf_counter()

            def mark(what, wi):
                if trace is not None:
                    trace.append((what, wi, (time.perf_counter() - t_begin) * 1e3))

            try:
                for wi in range(len(windows)):
                    window = [f.result() for f in pending]
                    mark("created", wi)
                    enqueue_many(window)      # queued behind (and beside) the previous window: the device never waits for the host
                    mark("enqueued", wi)
                    pending = ((all_pending[wi + 1] if eager_create else submit_window(windows[wi + 1])) if wi + 1 < len(windows) else [])   # built while the device solves
                    if windows_on_device < 2:
                        collect(window)
                        mark("collected", wi)
                        continue
                    if in_flight is not None:
                        done, in_flight = in_flight, None
                        collect(done)
                        mark("collected", wi - 1)
                    in_flight = window
                if in_flight is not None:
                    done, in_flight = in_flight, None
                    collect(done)
            finally:
                if in_flight is not None:     # a create or an enqueue raised while a window was still on the device: collect it (its tables hold
                    try:                      # streams and arena blocks) before the exception leaves
                        wait_many(in_flight)
                    except Exception:  # noqa: BLE001 -- the first error is the one to report
                        pass
                    for t in in_flight:
                        t.close()
            for f in releases:
                f.result()
        return tables

    import threading

    devices = list(devices)
    visible = device_count()
    if not devices or any(d < 0 or d >= visible for d in devices):
        raise RuntimeError(f"solve_blocks: devices {devices} requested but {visible} HIP device(s) visible")
    if weights is None:
        weights = [float(p.var_position.size) for p in problems]
    shares = assign_blocks(weights, len(devices))
    out = [None] * len(problems)
    errors = []

    def worker(slot):
        try:
            share = shares[slot]
            # every device worker builds its own tables: the host's create threads are shared out between them
            solved = solve_blocks([problems[b] for b in share], device=devices[slot], path=path,
                                  max_in_flight=max_in_flight, release=release, create_threads=max(1, create_threads // len(devices)),
                                  host_threads_per_create=host_threads_per_create, windows_on_device=windows_on_device,
                                  eager_create=eager_create, trace=trace)
            for b, t in zip(share, solved):
                out[b] = t
        except BaseException as exc:  # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(slot,), name=f"whamd-dev{devices[slot]}-{slot}") for slot in range(len(devices))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out


def merge_block_solutions(n_reads: int, n_individuals: int, blocks, solutions: Dict[int, dict]) -> dict:
    """Concatenates per-block solutions (dicts as produced by tests/helpers.table_solution) into the
    solution of the whole instance."""
    merged = {"cost": 0, "index_path": [], "transmission": [], "path_transmission": [], "positions": [],
              "partitioning": [1] * n_reads, "allele0": [[] for _ in range(n_individuals)],
              "allele1": [[] for _ in range(n_individuals)], "quality": [[] for _ in range(n_individuals)]}
    for b, (_, reads, _) in enumerate(blocks):
        s = solutions[b]
        merged["cost"] += s["cost"]
        for key in ("index_path", "transmission", "path_transmission", "positions"):
            merged[key].extend(s[key])
        for i in range(n_individuals):
            for key in ("allele0", "allele1", "quality"):
                merged[key][i].extend(s[key][i])
        for local, r in enumerate(reads):
            merged["partitioning"][int(r)] = s["partitioning"][local]
        merged["sample_ids"] = s["sample_ids"]
    return merged
