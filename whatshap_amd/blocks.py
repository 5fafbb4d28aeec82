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

import contextlib

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


_CREATE_POOLS = {}


@contextlib.contextmanager
def _create_pool(n_workers: int):
    """The create workers of ``solve_blocks``, kept between calls (starting and joining sixteen Python threads per call was 2 - 3 ms of a 50 ms step)."""
    from concurrent.futures import ThreadPoolExecutor

    pool = _CREATE_POOLS.get(n_workers)
    if pool is None:
        pool = _CREATE_POOLS[n_workers] = ThreadPoolExecutor(max_workers=n_workers, thread_name_prefix="whamd-create")
    yield pool


def solve_blocks(problems: Sequence[ProblemArrays], device: int = 0, path=None, max_in_flight: int = 8,
                 release: bool = True, devices: Sequence[int] = None, weights: Sequence[float] = None, create_threads: int = None,
                 windows_on_device: int = 1, trace: list = None, eager_create: bool = False, host_threads_per_create: int = None):
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
    default, less where the process's CPU budget -- ``host_cpu_budget`` -- is smaller; bench.py also tries one worker per table with four threads each and reports what it used).  ``trace``: a list that receives
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
    if create_threads is None or host_threads_per_create is None:
        # 16 workers x 2 threads where the host has them.  (``cpu_quota``: the boxes this was measured on grant a job 16 CPUs of time per 100 ms period with 256 in its
        # mask; as many single-threaded workers as the quota has CPUs was tried as the default for calls whose creates exceed a period's budget -- 96 coverage-15
        # tables: 154 - 156 ms per step, every step, against 121 - 170 ms for the bursts, 145 on average -- and not kept.)
        import os

        n_cpus = len(os.sched_getaffinity(0))
        if create_threads is None:
            create_threads = max(1, min(16, n_cpus))
        if host_threads_per_create is None:
            host_threads_per_create = max(1, min(2, n_cpus // max(1, create_threads)))
    if devices is None:
        # create (flatten + plan + upload: host work, the C library releases the GIL) of the NEXT window runs on a few threads while the
        # device solves the current one; the windows' host-side result extraction runs in parallel as well (wait_many)
        from concurrent.futures import ThreadPoolExecutor

        windows = [problems[start:start + max_in_flight] for start in range(0, len(problems), max_in_flight)]
        tables = []
        n_workers = max(1, min(create_threads, max_in_flight))
        with _create_pool(n_workers) as pool:
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

            window = None                 # the window being enqueued / collected right now
            ok = False
            try:
                for wi in range(len(windows)):
                    window = [f.result() for f in pending]
                    pending = []
                    mark("created", wi)
                    enqueue_many(window)      # queued behind (and beside) the previous window: the device never waits for the host
                    mark("enqueued", wi)
                    pending = ((all_pending[wi + 1] if eager_create else submit_window(windows[wi + 1])) if wi + 1 < len(windows) else [])   # built while the device solves
                    if windows_on_device < 2:
                        collect(window)
                        window = None
                        mark("collected", wi)
                        continue
                    if in_flight is not None:
                        done, in_flight = in_flight, None
                        collect(done)
                        mark("collected", wi - 1)
                    in_flight, window = window, None
                if in_flight is not None:
                    done, in_flight = in_flight, None
                    collect(done)
                ok = True
            finally:
                if not ok:
                    # a create, an enqueue or a collect raised: nothing may stay on the device or in the pool behind the exception -- the window in
                    # flight, the window that was being enqueued / collected and every table whose create is still running hold streams and arena
                    # blocks.  Secondary errors are swallowed: the first one is the one to report.
                    leftovers = list(in_flight or []) + [t for t in (window or []) if t not in (in_flight or [])]
                    try:
                        if leftovers:
                            wait_many(leftovers)
                    except Exception:  # noqa: BLE001
                        pass
                    futures = list(pending) + ([f for w in all_pending for f in w] if eager_create and all_pending else [])
                    for f in futures:
                        try:
                            t = f.result()
                        except Exception:  # noqa: BLE001
                            continue
                        if t not in tables and t not in leftovers:
                            leftovers.append(t)
                    for t in leftovers:
                        if t in tables:
                            continue
                        try:
                            t.close()
                        except Exception:  # noqa: BLE001
                            pass
            for f in releases:
                f.result()
            mark("released", len(windows))
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


# ------------------------------------------------------------------------------------------------ host CPUs of a rank
def parse_cpulist(text: str) -> List[int]:
    """"0-63,128-191" (sysfs cpulist) -> sorted CPU numbers."""
    out = []
    for tok in text.replace("\n", "").split(","):
        tok = tok.strip()
        if not tok:
            continue
        lo, _, hi = tok.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(out))


def rank_cpu_slices(n_ranks: int, allowed: Sequence[int], node_of_rank: Sequence[int] = None, node_cpus: Dict[int, Sequence[int]] = None,
                    core_of_cpu: Dict[int, int] = None) -> List[List[int]]:
    """The CPUs rank r's host threads (creates, result extraction) are bound to -- pure function of the topology, one entry per rank.

    With ``node_of_rank`` (NUMA node of rank r's GPU) and ``node_cpus`` (CPUs of a node): the ranks whose GPUs hang off one node share THAT
    node's CPUs evenly, in rank order; a rank whose node is unknown (-1) or has no allowed CPU takes part in an even split of the CPUs
    no node-bound rank uses (all allowed CPUs when nobody is node-bound).  Hardware threads of one core stay together (``core_of_cpu``:
    CPUs are ordered by core before they are cut, so two ranks never share a core unless there are more ranks than cores).  More ranks
    than CPUs in a share (oversubscribed tests): the CPUs are dealt out round-robin -- never an empty slice."""
    allowed = sorted(set(int(c) for c in allowed))
    if not allowed or n_ranks <= 0:
        return [[] for _ in range(max(n_ranks, 0))]
    core_of_cpu = core_of_cpu or {}
    def by_core(cpus):
        return sorted(cpus, key=lambda c: (core_of_cpu.get(c, c), c))
    def deal(cpus, ranks, out):
        cpus = by_core(cpus)
        k = len(ranks)
        if len(cpus) >= k:
            # whole cores where possible: cut at core boundaries nearest to the even split
            cores = []
            for c in cpus:
                key = core_of_cpu.get(c, c)
                if cores and cores[-1][0] == key:
                    cores[-1][1].append(c)
                else:
                    cores.append((key, [c]))
            units = cores if len(cores) >= k else [(c, [c]) for c in cpus]
            for i, r in enumerate(ranks):
                lo, hi = len(units) * i // k, len(units) * (i + 1) // k
                out[r] = sorted(c for _, group in units[lo:hi] for c in group)
        else:
            for i, r in enumerate(ranks):
                out[r] = [cpus[i % len(cpus)]]
    out = [None] * n_ranks
    bound = {}
    if node_of_rank is not None and node_cpus:
        for r in range(n_ranks):
            node = node_of_rank[r] if r < len(node_of_rank) else -1
            share = [c for c in node_cpus.get(node, ()) if c in set(allowed)] if node is not None and node >= 0 else []
            if share:
                bound.setdefault(node, []).append(r)
        for node, ranks in bound.items():
            deal([c for c in node_cpus[node] if c in set(allowed)], ranks, out)
    rest = [r for r in range(n_ranks) if out[r] is None]
    if rest:
        used = set(c for cpus in out if cpus for c in cpus)
        free = [c for c in allowed if c not in used] or allowed
        deal(free, rest, out)
    return out


def _sysfs(path: str):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def machine_topology():
    """(node -> CPUs, CPU -> core key) from sysfs; empty dicts where the files are absent."""
    import glob
    import os

    node_cpus = {}
    for path in sorted(glob.glob("/sys/devices/system/node/node[0-9]*/cpulist")):
        text = _sysfs(path)
        if text is not None:
            node_cpus[int(os.path.basename(os.path.dirname(path))[4:])] = parse_cpulist(text)
    core_of_cpu = {}
    for path in glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology/thread_siblings_list"):
        text = _sysfs(path)
        if text:
            cpu = int(path.split("/cpu/cpu")[1].split("/")[0])
            core_of_cpu[cpu] = min(parse_cpulist(text))
    return node_cpus, core_of_cpu


def device_numa_node(device: int) -> int:
    """NUMA node of HIP device ``device`` (sysfs ``numa_node`` of its PCI function, found through ``whamd_device_pci_bus_id``); -1 if unknown."""
    from . import _native

    try:
        bus = _native.device_pci_bus_id(device)
    except Exception:  # noqa: BLE001 -- no device, an older library: unknown
        return -1
    text = _sysfs(f"/sys/bus/pci/devices/{bus}/numa_node")
    try:
        return int(text) if text is not None else -1
    except ValueError:
        return -1


def cpu_quota(root: str = "/sys/fs/cgroup", self_cgroup: str = "/proc/self/cgroup") -> float:
    """CPU time the process's control group may use, in CPUs (cgroup v2 ``cpu.max`` "quota period", v1 ``cpu.cfs_quota_us`` / ``cpu.cfs_period_us``);
    0.0 if unlimited or unknown.  The MI355X boxes of this project give a job all 256 hardware
    threads in its affinity mask and a quota of 16 CPUs -- more busy threads than that and the whole process is frozen for the rest of every 100 ms period."""
    import os

    best = 0.0
    dirs = [root]
    try:
        with open(self_cgroup) as f:
            for line in f:
                line = line.strip()
                if line.startswith("0::") and len(line) > 4:
                    dirs.append(root + line[3:])
    except OSError:
        pass
    for d in dirs:
        try:
            with open(os.path.join(d, "cpu.max")) as f:
                words = f.read().split()
            if len(words) == 2 and words[0] != "max" and float(words[1]) > 0:
                cpus = float(words[0]) / float(words[1])
                best = cpus if best == 0.0 else min(best, cpus)
        except (OSError, ValueError):
            pass
    try:
        with open(os.path.join(root, "cpu", "cpu.cfs_quota_us")) as f:
            quota = float(f.read().split()[0])
        with open(os.path.join(root, "cpu", "cpu.cfs_period_us")) as f:
            period = float(f.read().split()[0])
        if quota > 0 and period > 0:
            best = quota / period if best == 0.0 else min(best, quota / period)
    except (OSError, ValueError, IndexError):
        pass
    return best


def host_cpu_budget(n_cpus: int, local_world: int = 1, quota: float = None) -> int:
    """Threads one rank should keep busy: its CPU slice, capped by its share of the control group's quota (``cpu_quota``; the ranks of a node live in one group)."""
    quota = cpu_quota() if quota is None else quota
    if quota and quota > 0:
        return max(1, min(int(n_cpus), int(quota // max(1, local_world)) or 1))
    return max(1, int(n_cpus))


def bind_rank_to_device_cpus(local_rank: int, local_world: int, devices: Sequence[int] = None, apply: bool = True, spread_nodes: bool = False) -> dict:
    """One process per GPU: keeps rank ``local_rank``'s host threads -- the creates of ``whamd_dptable_create`` (csrc/host_parallel.h binds its
    workers inside the process's affinity mask), ``wait_many``'s result extraction, Python's own worker threads -- on the CPUs next to ITS GPU:
    the NUMA node of device ``devices[local_rank]`` (default: device = local rank), shared evenly with the other ranks whose GPUs hang off the
    same node; where the node is unknown, an even split of ``sched_getaffinity``.  Without this eight torchrun ranks pile onto whatever socket the
    scheduler picks and every rank's workers bind to "the node the first caller ran on" (host_parallel.h).  Call BEFORE the first create.
    No reference counterpart (the reference has no threads): this is the host half of the north star's "near-linear scaling to 8 GPUs".
    ``spread_nodes``: ignore where the visible devices sit and spread the ranks evenly over the NUMA nodes (what a full 8-GPU node looks like).
    Returns what it did: ``{"cpus": [...], "node": n, "source": "numa" | "even split", "applied": bool}``; ``WHAMD_NO_AFFINITY=1`` disables it."""
    import os

    allowed = sorted(os.sched_getaffinity(0))
    devices = list(range(local_world)) if devices is None else list(devices)
    node_cpus, core_of_cpu = machine_topology()
    node_of_rank = [device_numa_node(d) for d in devices] if node_cpus else None
    if spread_nodes and node_cpus:
        # a dry run of "rank r of n" on a box with fewer GPUs than ranks (bench.py's create_rate): the GPUs of a full node are spread evenly over its
        # NUMA nodes, so the ranks are too
        nodes = sorted(node_cpus)
        node_of_rank = [nodes[r * len(nodes) // local_world] for r in range(local_world)]
    slices = rank_cpu_slices(local_world, allowed, node_of_rank, node_cpus, core_of_cpu)
    mine = slices[local_rank] if 0 <= local_rank < len(slices) else []
    node = node_of_rank[local_rank] if node_of_rank and local_rank < len(node_of_rank) else -1
    quota = cpu_quota()
    info = {"cpus": mine, "n_cpus": len(mine), "node": node, "source": "numa" if node is not None and node >= 0 and node_cpus.get(node) else "even split",
            "applied": False, "cpu_quota": quota, "cpu_budget": host_cpu_budget(len(mine) or len(allowed), local_world, quota)}
    if apply and mine and not os.environ.get("WHAMD_NO_AFFINITY"):
        try:
            os.sched_setaffinity(0, mine)
            info["applied"] = True
        except OSError as exc:
            info["error"] = str(exc)
    return info


_CLOSE_POOL = None


def close_tables(tables, threads: int = 8) -> None:
    """``whamd_dptable_destroy`` of many tables on a few threads: a destroy hands ~100 host blocks back to the library's pool (60 - 70 us per
    coverage-15 table, 7 ms for the 96 tables of one step when done one after the other); the C library is thread-safe and ctypes releases the GIL."""
    global _CLOSE_POOL
    tables = list(tables)
    if len(tables) < 4 or threads <= 1:
        for t in tables:
            t.close()
        return
    if _CLOSE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor

        _CLOSE_POOL = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="whamd-close")
    n = min(threads, len(tables))
    list(_CLOSE_POOL.map(lambda part: [t.close() for t in part], [tables[i::n] for i in range(n)]))


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
