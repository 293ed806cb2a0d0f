#!/usr/bin/env python3
"""bench.py -- GCUPS of the DP-fill hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (seqalign_fill_batch_device: the fill of the
match / gap_a / gap_b matrices, reference src/alignment.c:28-168) over one
synthetic batch that is already resident in HBM.

  N = 1 : BASELINE configs[1] (C2) -- 10 000 NW pairs, DNA 150x150, 1/-2/-4/-1.
  N > 1 : BASELINE configs[4] (C5) -- 1 M NW pairs, DNA 150x150, seed 5, sharded by
          contiguous pair index: rank g fills pairs [g*125 000, (g+1)*125 000) of the
          1 M-pair stream (34 GB of matrices per GPU).  Per-GPU work is fixed, so
          `scaling` is "weak"; at N = 8 the job is exactly C5.  `--scaling strong`
          shards the whole 1 M pairs over N ranks instead (N >= 2: a shard must fit HBM).
          Pairs are independent: NO collective on the data path.  The control plane
          (barrier, MAX of the elapsed time, SUM of the cells, per-rank figures for the report) is a
          handful of scalars over a key-value store (torch.distributed.TCPStore: the launcher's own
          when run under torch.distributed.run) -- no process group, no RCCL
          (SEQALIGN_DIST_BACKEND=gloo|nccl switches to torch.distributed collectives).
          Each rank pins itself (and the library's host worker threads, which inherit the mask) to its
          share of the CPUs of its GPU's NUMA node.

    python bench.py --gpus 1 --steps 1000 --warmup 10
    python bench.py --gpus 8                     # launches its 8 ranks itself
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8   # or under a launcher

Prints ONE JSON line (rank 0).  `value` = all ranks' cells / max-rank seconds.
`roofline`    : algorithmic bytes per launch / mean kernel duration (HIP events on
                the launch stream, recorded inside the timed region) vs 8 TB/s.
`e2e`         : wall clock of the host-level call on the same batch (host buffers in ->
                H2D -> fill -> device traceback -> strings out), PCIe inclusive, median of 5 calls
                after 0.25 s of the same call (its own steady state); never `value`.
`cpu_baseline`: the reference itself (oracle/_ref, built from /root/reference in the
                authoring container) or, if absent, our C restatement (oracle/), timed
                on this box's host cores by a pthread harness (oracle/cpu_bench.c) on a
                bounded sample: 1 pinned thread, and as many threads as the container is granted
                (min of physical cores, affinity mask, cgroup CPU quota -- all three in the line,
                with the all-cores / (one thread x threads) ratio); fill only, and fill + traceback
                (`cpu_baseline.e2e`: needleman_wunsch_align2 of the compiled reference) beside `e2e`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for p in (ROOT / "seq-align_amd" / "python", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

from seqalign_amd import workloads as W      # noqa: E402  (numpy only; the HIP library loads in run())

HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
C5_TOTAL_PAIRS = 1_000_000
C5_RANKS = 8

WORKLOADS = {
    # name: (generator, kwargs, pairs per GPU, is_sw, scoring spec, description)
    "C2": ("dna_nw_150", dict(seed=1), 10000, 0, {"preset": "default"},
           "10k NW pairs, DNA 150x150, default scoring 1/-2/-4/-1 (BASELINE configs[1])"),
    "C3": ("dna_sw_read_vs_ref", dict(seed=2), 10000, 1, {"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]},
           "10k SW pairs, DNA 150x1000 read-vs-ref, 2/-2/-2/-1 (BASELINE configs[2])"),
    "L1000": ("dna_nw_150", dict(seed=4, length=1000), 1000, 0, {"preset": "default"},
              "1k NW pairs, DNA 1000x1000 (long-sequence check, not a BASELINE config)"),
    "C4": ("protein_sw_300", dict(seed=3), 4000, 1, {"preset": "BLOSUM62"},
           "4k SW pairs, protein 300x300, BLOSUM62 (BASELINE configs[3])"),
    "C5": ("dna_nw_indexed", dict(seed=5), C5_TOTAL_PAIRS // C5_RANKS, 0, {"preset": "default"},
           "1M NW pairs, DNA 150x150, seed 5, sharded over 8 GPUs: 125k pairs (34 GB of matrices) per GPU "
           "(BASELINE configs[4])"),
}


# ------------------------------------------------------------------ launcher ---
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` outside any launcher: start the N ranks ourselves, one process per
    GPU, same environment contract as torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Rank 0 owns stdout (the one JSON line); the others only stderr."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + float(os.environ.get("SEQALIGN_BENCH_TIMEOUT", "3000"))
    for p in procs:
        try:
            rc = max(rc, abs(p.wait(timeout=max(1.0, deadline - time.time()))))
        except subprocess.TimeoutExpired:
            rc = max(rc, 124)
    for p in procs:           # a rank that died leaves the others in the barrier: kill exactly our children
        if p.poll() is None:
            p.kill()
    return rc


def _cpulist(text: str) -> list[int]:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_cpus(device: int, sysfs: Path = Path("/sys")):
    """(numa node, its CPUs) of HIP device `device`, from the PCI address torch reports; (None, []) if unknown."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        dev = sysfs / "bus" / "pci" / "devices" / bdf
        node = int((dev / "numa_node").read_text())
        cpus = _cpulist((dev / "local_cpulist").read_text())
        if node < 0:
            return None, []
        return node, cpus
    except Exception:
        return None, []


def cgroup_cpu_quota():
    """CPUs the container's cgroup grants (cpu.max / cfs_quota_us), or None: what the scheduler enforces whatever the
    affinity mask shows (the 1-GPU development boxes show 256 CPUs and grant 16)."""
    try:
        q, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        return None if q == "max" else max(1, int(q) // int(period))
    except (OSError, ValueError):
        pass
    try:
        q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        return None if q <= 0 else max(1, q // period)
    except (OSError, ValueError):
        return None


def pin_rank(node, cpus, local_rank: int, local_world: int, nodes_of_ranks: list) -> dict:
    """Pin this process to its share of its GPU's NUMA node: the node's CPUs are dealt out evenly among the local
    ranks whose GPUs hang off the same node, in rank order (hyper-thread siblings -- cpu c and c + n/2 on this
    platform -- end up in the same share because the list is cut into contiguous runs of physical ids).
    The library's worker pool is created later and inherits the mask (its size = CPUs in the mask, <= 32)."""
    if node is None or not cpus or not hasattr(os, "sched_setaffinity"):
        return {"numa_node": node, "cpus": None}
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    peers = [r for r in range(local_world) if nodes_of_ranks[r] == node]
    if not allowed or local_rank not in peers:
        return {"numa_node": node, "cpus": None}
    k, m = peers.index(local_rank), len(peers)
    half = len(allowed) // 2
    if half and all(b - a == 1 for a, b in zip(allowed[:half], allowed[1:half])) and len(allowed) % 2 == 0 and len(allowed) >= 2 * m:
        # two runs (first threads, then their siblings): take the same slice of both
        a, b = allowed[:half], allowed[half:]
        per = half // m
        mine = a[k * per:(k + 1) * per] + b[k * per:(k + 1) * per]
    else:
        per = max(1, len(allowed) // m)
        mine = allowed[k * per:(k + 1) * per] or allowed
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return {"numa_node": node, "cpus": None}
    return {"numa_node": node, "cpus": len(mine), "first_cpu": mine[0]}


def make_shard(name: str, rank: int, world: int, pairs: int, scaling: str):
    """(batch of THIS rank, global pair count, first global pair index)."""
    gen, kwargs, per_gpu, *_ = WORKLOADS[name]
    if name == "C5":
        total = (pairs * world) if pairs else (C5_TOTAL_PAIRS if scaling == "strong" else per_gpu * world)
        lo, hi = W.shard_range(total, rank, world)
        return W.dna_nw_indexed(lo, hi - lo, **kwargs), total, lo
    per_gpu = pairs or per_gpu
    total = per_gpu * world
    full = getattr(W, gen)(total, **kwargs)
    lo, hi = W.shard_range_cells(full.matrix_cells(), rank, world)
    return full.slice(lo, hi), total, lo


# -------------------------------------------------------------- cpu baseline ---
def host_cpu_info() -> dict:
    info = {"logical_cpus": os.cpu_count() or 1, "model": None, "physical_cores": None, "governor": None,
            "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    try:
        cores, phys, core = set(), None, None
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and not info["model"]:
                info["model"] = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
        info["physical_cores"] = len(cores) or None
    except OSError:
        pass
    try:
        info["governor"] = Path("/sys/devices/system/cpu/cpu0/cpufreq/scaling_governor").read_text().strip()
    except OSError:
        info["governor"] = "unknown (no cpufreq sysfs)"
    return info


def cpu_threads(cpu: dict) -> tuple[int, dict]:
    """Threads the all-cores leg may use = min(physical cores, CPUs in the affinity mask, the cgroup's CPU quota):
    what the scheduler will actually run side by side.  (Rounds 1-4 started one thread per physical core the box SHOWS --
    128 -- on a 16-CPU quota and called the result "128 cores".)"""
    limits = {"physical_cores": cpu.get("physical_cores"), "affinity_cpus": cpu.get("affinity_cpus"),
              "cgroup_cpu_quota": cgroup_cpu_quota()}
    known = [v for v in limits.values() if v]
    return (min(known) if known else (cpu.get("logical_cpus") or 1)), limits


def cpu_baseline(batch, spec, is_sw, budget_s=20.0, min_score=0):
    """Reference CPU path on THIS host, bounded sample: fill only (the metric) and fill + traceback (beside `e2e`), each on one
    pinned thread and on every core the container is granted.  Checker code (oracle/): allowed here, and only here plus the
    out-of-timed-region spot check."""
    import orclib as O
    lib_path = ROOT / "oracle" / "libcpubench.so"
    src = ROOT / "oracle" / "cpu_bench.c"
    if not lib_path.exists() or (src.exists() and lib_path.stat().st_mtime < src.stat().st_mtime):
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "libcpubench.so"], check=True, stdout=subprocess.DEVNULL)
    hb = C.CDLL(str(lib_path))
    hb.cpubench_run_aux.restype = C.c_double
    ref, orc = O.ref(), O.oracle()
    vp = lambda f: C.cast(f, C.c_void_p)
    import seqalign_amd
    osc = O.Scoring.from_buffer_copy(bytes(seqalign_amd.make_scoring(spec)))
    if ref is not None:
        kind, mode = "reference", 0
        sc = O.build_scoring(spec, "ref")
        fn, destroy = vp(ref.aligner_align), vp(ref.aligner_destroy)
        what = "aligner_align of the compiled reference (oracle/_ref, src/alignment.c:170-193)"
    else:
        kind, mode, sc = "port", 1, osc
        fn, destroy = vp(orc.orc_fill), C.c_void_p(0)
        what = "orc_fill of the C restatement (oracle/seqalign_oracle.c)"
    # fill + traceback: the reference's own NW front-end when it was compiled; the restatement for SW (smith_waterman.c
    # needs the un-vendored sort_r: not in _ref) and wherever _ref is absent
    aux_t = C.c_void_p * 4
    if not is_sw and ref is not None:
        ref.needleman_wunsch_new.restype = C.c_void_p
        ref.alignment_create.restype = C.c_void_p
        e_kind, e_mode, e_sc, e_fn = "reference", 2, sc, vp(ref.needleman_wunsch_align2)
        e_aux = aux_t(vp(ref.needleman_wunsch_new).value, vp(ref.needleman_wunsch_free).value,
                      vp(ref.alignment_create).value, vp(ref.alignment_free).value)
        e_what = ("needleman_wunsch_align2 of the compiled reference (src/needleman_wunsch.c:34-146): fill + end-cell pick + "
                  "traceback into an alignment_t")
    elif not is_sw:
        e_kind, e_mode, e_sc, e_fn, e_aux = "port", 3, osc, vp(orc.orc_nw_align), aux_t()
        e_what = "orc_nw_align of the C restatement: fill + traceback into strings"
    else:
        e_kind, e_mode, e_sc, e_fn = "port", 4, osc, vp(orc.orc_fill)
        e_aux = aux_t(vp(orc.orc_sw_hits).value, None, None, None)
        e_what = (f"orc_fill + orc_sw_hits(min_score={min_score}, max_hits=1) of the C restatement (the reference's SW front-end, "
                  "src/smith_waterman.c:137-277, needs the absent sort_r and is not in oracle/_ref): fill + candidate sort + best hit")
    n = min(batch.n_pairs, 20000)     # the sample the threads cycle over

    def run(fn_, destroy_, aux_, mode_, sc_, threads, seconds):
        cells, pairs = C.c_uint64(0), C.c_uint64(0)
        dt = hb.cpubench_run_aux(fn_, destroy_, aux_, C.c_int(mode_), C.byref(sc_), batch.arena.ctypes.data_as(C.c_char_p),
                                 batch.off_a.ctypes.data_as(C.c_void_p), batch.len_a.ctypes.data_as(C.c_void_p),
                                 batch.off_b.ctypes.data_as(C.c_void_p), batch.len_b.ctypes.data_as(C.c_void_p),
                                 C.c_size_t(n), C.c_int(is_sw), C.c_int32(min_score), C.c_int(threads), C.c_double(seconds),
                                 C.byref(cells), C.byref(pairs))
        if dt <= 0:
            raise RuntimeError("cpubench_run_aux failed")
        return cells.value / dt / 1e9, pairs.value, dt

    cpu = host_cpu_info()
    threads, limits = cpu_threads(cpu)
    cpu["cgroup_cpu_quota"] = limits["cgroup_cpu_quota"]
    how = ("one per physical core" if threads == limits["physical_cores"] else
           f"the container's CPU grant: min of physical cores {limits['physical_cores']}, affinity mask {limits['affinity_cpus']}, "
           f"cgroup CPU quota {limits['cgroup_cpu_quota']}")

    def leg(fn_, destroy_, aux_, mode_, sc_, kind_, what_, t1, tn, note):
        v1, p1, d1 = run(fn_, destroy_, aux_, mode_, sc_, 1, t1)
        out = dict(value=v1, unit="GCUPS", cores=1, kind=kind_,
                   sample=f"{p1} pairs through {what_}, {note}, 1 pinned thread, {d1:.1f} s (pthread harness oracle/cpu_bench.c)")
        vn, pn, dn = run(fn_, destroy_, aux_, mode_, sc_, threads, tn)
        out["all_cores"] = dict(value=vn, unit="GCUPS", cores=threads, threads=threads,
                                cgroup_cpu_quota=limits["cgroup_cpu_quota"],
                                scaling_vs_one_thread=vn / (v1 * threads) if v1 > 0 else None,
                                sample=f"{pn} pairs, one aligner per thread, {threads} pinned threads ({how}), {dn:.1f} s")
        return out

    out = leg(fn, destroy, None, mode, sc, kind, what, budget_s * 0.30, budget_s * 0.25, "fill only")
    out["host"] = cpu
    out["e2e"] = leg(e_fn, C.c_void_p(0), e_aux, e_mode, e_sc, e_kind, e_what, budget_s * 0.25, budget_s * 0.20,
                     "fill + traceback")
    return out


# ----------------------------------------------------------------- plumbing ---
def plumbing_test(args, rank, world, grp):
    """CPU-only self-test of the multi-rank PLUMBING (launcher, sharding, reductions) -- NOT a measurement
    and not the product: the oracle fills a handful of pairs so that tests/test_multigpu_cpu.py can check that
    the ranks' shards concatenate to the single-process result.  Prints a line that says so."""
    import orclib as O
    batch, total, first = make_shard(args.workload, rank, world, args.pairs or 5, args.scaling)
    sc = O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
    digests = []
    for p in range(batch.n_pairs):
        rc, M, A, B = O.oracle_fill(sc, batch.seq_a(p), batch.seq_b(p), 0)
        digests.append([first + p, O.fnv(M), O.fnv(A), O.fnv(B)])
    grp.barrier()
    elapsed = grp.max_float(1.0 + rank)
    cells = grp.sum_int(batch.cells())
    gathered = grp.gather_objects(digests)
    if rank == 0:
        print(json.dumps({"metric": "PLUMBING_TEST_NOT_A_MEASUREMENT", "n_gpus": world, "global_pairs": total,
                          "elapsed_max": elapsed, "cells_sum": cells,
                          "digests": [d for part in gathered for d in part]}), flush=True)
    grp.close()
    return 0


# ----------------------------------------------------------------------- run ---
def run(args) -> int:
    from seqalign_amd.dist import Group, env_world

    rank, local, world = env_world()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    args.default_workload = args.workload is None
    workload = args.workload or ("C2" if world == 1 else "C5")
    args.workload = workload
    backend = os.environ.get("SEQALIGN_DIST_BACKEND", "store")
    if args.plumbing_test:
        return plumbing_test(args, rank, world, Group(backend if backend != "nccl" else "store"))

    import torch
    import seqalign_amd as S
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the product has no CPU path")
    n_dev = torch.cuda.device_count()
    folded = world > n_dev            # fewer GPUs than ranks (1-GPU development box): ranks share devices
    local = local % n_dev
    torch.cuda.set_device(local)
    grp = Group(backend, torch.device("cuda", local) if backend == "nccl" else None)
    # NUMA: this rank and the library's host threads onto the CPUs next to this rank's GPU
    node, node_cpus = gpu_numa_cpus(local)
    pin = {"numa_node": node, "cpus": None}
    if world > 1 and not args.no_pin:
        nodes = grp.gather_objects(node)
        pin = pin_rank(node, node_cpus, rank, world, nodes)
    if world > 1 and "SEQALIGN_HOST_THREADS" not in os.environ:
        # the library sizes its worker pool to min(CPUs in the mask, the cgroup's CPU quota, 32) -- per PROCESS; N ranks in one
        # container share the quota, so each gets its share of it (the pool is created at the first host-level call)
        quota = cgroup_cpu_quota()
        share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        if quota:
            share = min(share, max(2, quota // world))
        os.environ["SEQALIGN_HOST_THREADS"] = str(max(1, min(32, share)))
        pin["host_threads"] = int(os.environ["SEQALIGN_HOST_THREADS"])
        pin["cgroup_cpu_quota"] = quota

    S.lib()                                          # raises if the HIP library is missing: no fallback
    env = dict(rank=rank, local=local, world=world, backend=backend, folded=folded, pin=pin)
    out = measure(args, grp, env, workload, args.steps, args.warmup, args.pairs)
    if rank == 0:
        # Every other BASELINE config in the SAME driver-run line: configs[2] (C3), configs[3] (C4) and one GPU's
        # share of configs[4] (C5: 125 000 of the 1 M pairs), each through the same measurement as the headline -- its own context,
        # its own placed arenas, the fill timed with HIP events on the launch stream, the same kernel on unplaced arenas, the
        # separate SW reduction, the host-level calls -- a few seconds each.  `value` stays C2's.
        if world == 1 and args.default_workload and not args.no_configs:
            out["configs"] = {}
            for name, steps in (("C3", 30), ("C4", 100), ("C5", 20)):
                t_cfg = time.perf_counter()
                try:
                    r = measure(args, grp, env, name, steps, min(args.warmup, 5), 0)
                    entry = {"workload": r["config"]["workload"], "pairs": r["config"]["pairs_per_gpu"], "value": r["value"],
                             "unit": "GCUPS", "steps": steps, "ms_per_step": r["ms_per_step"], "kernel": r["config"]["kernel"],
                             "bit_exact_vs_oracle": r["bit_exact_vs_oracle"], "roofline": r["roofline"],
                             "arena_placement_quality": r["config"]["arena_placement_quality"],
                             "arena_placement_search": r["config"]["arena_placement_search"]}
                    for k in ("sw_reduce", "e2e"):
                        if k in r:
                            entry[k] = r[k]
                except Exception as ex:      # the headline must survive a failing extra: say what failed, in the line
                    entry = {"error": f"{type(ex).__name__}: {ex}"}
                entry["seconds_spent"] = round(time.perf_counter() - t_cfg, 2)
                out["configs"]["C5_share" if name == "C5" else name] = entry
        # north_star: the 1 / 2 / 4 / 8-GPU figures "next to the reference CPU path" -- rank 0 times it at every N (once, the same
        # bounded leg; the other ranks wait in the barrier below with their GPUs idle: it is outside every timed region)
        if not args.no_cpu_baseline:
            gen, kwargs, _, is_sw, spec, desc = WORKLOADS[workload]
            b0 = out.pop("_batch")
            sc0 = S.make_scoring(spec)
            out["cpu_baseline"] = cpu_baseline(b0, spec, is_sw, min_score=(
                W.default_minscore(sc0.match, int(b0.len_a[0]), int(b0.len_b[0])) if is_sw else 0))
        out.pop("_batch", None)
        print(json.dumps(out), flush=True)
    grp.barrier()
    grp.close()
    return 0


def measure(args, grp, env, workload, steps, warmup, pairs):
    """One workload through the whole measurement (fill roofline with placed and unplaced arenas, the SW reduction, the host-level
    calls): the headline's, and each entry of `configs`.  Returns the line's dict on rank 0 (None elsewhere)."""
    import torch
    import seqalign_amd as S
    rank, local, world, backend, folded, pin = (env[k] for k in ("rank", "local", "world", "backend", "folded", "pin"))
    gen, kwargs, _, is_sw, spec, desc = WORKLOADS[workload]
    batch, global_pairs, first_pair = make_shard(workload, rank, world, pairs, args.scaling)

    ctx = S.Context(local)
    sc = S.make_scoring(spec)
    h = ctx.upload_scoring(sc, is_sw)
    db = S.DeviceBatch(batch, local, placement=args.placement, ctx=ctx)   # arenas from seqalign_arenas_alloc
    t_placed = time.perf_counter()

    # `e2e` cold: the host-level call right after the placement, before anything else has run -- the first call (it sizes the
    # context's scratch buffers, uploads the scoring) and the median of the next five, no waiting
    e2e_cold = None
    if not args.no_e2e:
        if is_sw:
            thr0 = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
            cold_fn = lambda: ctx.sw_batch(batch, sc, thr0, max_hits=1, hit_cap=batch.n_pairs + 8, raw=True)
        else:
            cold_fn = lambda: ctx.nw_batch(batch, sc, raw=True)
        ts = []
        for _ in range(6):
            t1 = time.perf_counter()
            cold_fn()
            ts.append((time.perf_counter() - t1) * 1e3)
        e2e_cold = {"first_call_ms": ts[0], "cold_ms": float(np.median(ts[1:])),
                    "s_after_placement": time.perf_counter() - t_placed}

    # kernel choice: measured, not guessed -- among the kernels that have ever won a configuration (stream: rows up to 768
    # columns; wgstream: longer rows); --all-kernels also times the correctness paths (wavefront, rowscan, strips)
    if args.kernel == "auto":
        best = None
        candidates = ((S.KERNEL_WAVEFRONT, S.KERNEL_ROWSCAN, S.KERNEL_STREAM, S.KERNEL_STRIPS, S.KERNEL_WGSTREAM) if args.all_kernels
                      else (S.KERNEL_STREAM, S.KERNEL_WGSTREAM))
        for k in candidates:
            ms = db.time_fill_ms(ctx, h, k, 6)[1:]
            m = float(np.median(ms))
            if best is None or m < best[1]:
                best = (k, m)
        kernel = grp.broadcast_int(best[0], 0)   # all ranks run the same kernel
    else:
        kernel = {v: k for k, v in S.KERNEL_NAMES.items()}[args.kernel]

    total_cells = grp.sum_int(batch.cells())       # whole-job cells per step
    # (measured BEFORE the timed loop of the headline: right after a second of back-to-back HBM-bound launches the same
    # call measures 25 % slower -- C2 1.18 instead of 0.93 ms -- which says something about the GPU's clocks under
    # sustained load, not about the call)
    # end to end through the host-level entry point on the same batch (never `value`)
    e2e = None
    if not args.no_e2e:
        if is_sw:
            thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
            call, fn = "seqalign_sw_batch(max_hits=1)", lambda: ctx.sw_batch(batch, sc, thr, max_hits=1, hit_cap=batch.n_pairs + 8, raw=True)
        else:
            call, fn = "seqalign_nw_batch", lambda: ctx.nw_batch(batch, sc, raw=True)
        def settle(f):
            """The call in ITS OWN steady state: the first call sizes the context's scratch buffers, and the GPU's clocks
            take a while to follow a change of regime (after the HBM-bound launches of the kernel choice above the same
            call measured up to 40 % slower: C5's share 9.7 instead of 6.8 ms) -- repeat it for 0.25 s first."""
            t_end = time.perf_counter() + 0.25
            n_calls = 0
            while n_calls < 3 or time.perf_counter() < t_end:
                f()
                n_calls += 1
        # The arena placement above created and released up to 160 GiB of HBM chunks.  In round 3 every host-level call measured
        # ~30 % slower for 1.4-4.4 s after that (profiles/r03/r03_after_placement_transient.txt) and this loop waited 6 s; with
        # round 4's host path (moves home, no blit kernel, a worker pool sized to the CPU quota) the same experiment shows no
        # transient at all (tools/placement_pressure.py, profiles/r04/r04_placement_pressure.txt: 0.49 ms from the first 250 ms
        # after walks of 24, 64 and 160 GiB).  What remains is paid by the FIRST call when it allocates gigabytes right after a walk that
        # used its whole budget: the driver clears what the walk handed back (first_call_ms 4 s for C5 / C3 after 165 GiB, 20-75 ms
        # after shorter walks).  `cold_ms` above is measured without any wait; this keeps calling for a second.
        while time.perf_counter() - t_placed < args.e2e_after:
            fn()
        settle(fn)
        walls = []
        for _ in range(5):
            grp.barrier()
            t1 = time.perf_counter()
            fn()
            walls.append(time.perf_counter() - t1)
        wall_mine = float(np.median(walls))
        wall = grp.max_float(wall_mine)
        e2e = {"call": call, "ms": wall * 1e3, "value": total_cells / wall / 1e9, "unit": "GCUPS",
               "first_call_ms": grp.max_float(e2e_cold["first_call_ms"]), "cold_ms": grp.max_float(e2e_cold["cold_ms"]),
               "cold_measured_s_after_placement": round(e2e_cold["s_after_placement"], 2),
               "launched": ctx.last_call(),
               "includes": "host pack, H2D, fill, device traceback, results home, host expansion into the caller's strings",
               "path": ("plain scoring: the fill writes one byte of directions per cell (NW) / match_scores + that byte (SW, "
                        "max_hits > 1) instead of the three matrices, the walks follow the bytes (DESIGN.md 3.5b) and send home two "
                        "bits per alignment column, which the host expands (NW; DESIGN.md 3.5d); `value` / `roofline` above are "
                        "the three-matrix fill, the BASELINE metric"),
               "ms_this_rank": wall_mine * 1e3}
        if is_sw:   # the multi-hit path: reverse sweep + one traceback per hit
            fn4 = lambda: ctx.sw_batch(batch, sc, thr, max_hits=4, hit_cap=4 * batch.n_pairs + 8, raw=True)
            settle(fn4)
            walls = []
            for _ in range(5):
                grp.barrier()
                t1 = time.perf_counter()
                fn4()
                walls.append(time.perf_counter() - t1)
            wall4 = grp.max_float(float(np.median(walls)))
            e2e["up_to_4_hits"] = {"call": "seqalign_sw_batch(max_hits=4)", "ms": wall4 * 1e3,
                                   "value": total_cells / wall4 / 1e9, "unit": "GCUPS"}

    # ... and with CIGAR as its output (seqalign_nw_batch_cigar: run lengths straight from the walks' bit planes, no strings expanded;
    # north_star: "identical CIGAR/alignment strings"), same batch, same steady state
    if e2e is not None and not is_sw:
        fnc = lambda: ctx.nw_batch_cigar(batch, sc, 1, raw=True)      # (worst-case slots, 2 (len_a + len_b) + 2 bytes: random pairs' CIGARs are long)
        try:
            settle(fnc)
            walls = []
            for _ in range(5):
                grp.barrier()
                t1 = time.perf_counter()
                fnc()
                walls.append(time.perf_counter() - t1)
            wc = grp.max_float(float(np.median(walls)))
            e2e["cigar"] = {"call": "seqalign_nw_batch_cigar(SEQALIGN_CIGAR_M, worst-case slots)", "ms": wc * 1e3, "value": total_cells / wc / 1e9,
                            "unit": "GCUPS", "launched": ctx.last_call()}
        except S.SeqAlignError as ex:      # (a CIGAR longer than its 64-byte slot: say so, do not hide the line)
            e2e["cigar"] = {"error": str(ex)}

    # the same call as a STREAM of batches (seqalign_*_batch_submit / seqalign_job_wait, sa_async.hip): `in_flight` batches at
    # a time, the caller waiting that many behind its submits -- what an API caller streaming batches of this size gets, steady
    # state: batch k + 1's packing and upload run beside batch k's walk and expansion (never `value`)
    if e2e is not None and not args.no_stream:
        in_flight, n_stream = 3, args.stream_batches
        if is_sw:
            mk = lambda: ((S.SwHit * (batch.n_pairs + 8))(), *(np.zeros(min(int((batch.n_pairs + 8) * (int(batch.len_a.max()) + int(batch.len_b.max()) + 2)), 1 << 30), np.uint8) for _ in range(2)))
            sub = lambda buf: ctx.sw_batch_submit(batch, sc, thr, max_hits=1, hit_cap=batch.n_pairs + 8, buffers=buf)
            call_s = "seqalign_sw_batch_submit(max_hits=1)"
        else:
            mk = lambda: ctx.nw_buffers(batch)
            sub = lambda buf: ctx.nw_batch_submit(batch, sc, buf)
            call_s = "seqalign_nw_batch_submit"
        bufs = [mk() for _ in range(in_flight + 1)]
        for j in [sub(b) for b in bufs]:          # every lane sizes its scratch, every buffer is touched
            j.wait(raw=True)

        def stream_once(n_b):
            pending = []
            for k in range(n_b):
                if len(pending) == in_flight:
                    pending.pop(0).wait(raw=True)
                pending.append(sub(bufs[k % (in_flight + 1)]))
            for j in pending:
                j.wait(raw=True)
        stream_once(2 * in_flight)
        # (the lanes' host threads share the box's CPUs with whatever else runs there: one stream of 32 batches took 1.8 - 3.1 ms per C3
        #  batch within one minute on one box -- three streams, the median one is the figure, all three are in the line)
        dts = []
        for _ in range(3):
            grp.barrier()
            t1 = time.perf_counter()
            stream_once(n_stream)
            dts.append(grp.max_float(time.perf_counter() - t1))
        dt = sorted(dts)[1]
        e2e["stream"] = {"call": f"{call_s} x {n_stream}, {in_flight} in flight (seqalign_job_wait {in_flight} behind the submits), own output buffers per job in flight",
                         "batches": n_stream, "in_flight": in_flight, "ms_per_batch": dt / n_stream * 1e3,
                         "value": total_cells * n_stream / dt / 1e9, "unit": "GCUPS",
                         "vs_synchronous": (total_cells * n_stream / dt) / (total_cells / wall),
                         "streams_ms_per_batch": [round(x / n_stream * 1e3, 4) for x in dts], "of_three_streams": "the median"}

    torch.cuda.synchronize()
    for _ in range(warmup):
        db.fill(ctx, h, kernel, order_after_current=False)
    torch.cuda.synchronize()
    grp.barrier()

    # N > 1: what ONE GPU does on this workload with the others idle (rank 0, a few steps) -- the base the 1 -> N efficiency
    # should be read against: the N = 1 line of this file is C2 (10 k pairs), the N > 1 lines are C5's 125 k-pair shares
    scale_base = None
    if world > 1:
        if rank == 0:
            torch.cuda.synchronize()
            n_alone, t_a = 0, time.perf_counter()
            while n_alone < 3 or (time.perf_counter() - t_a < 0.25 and n_alone < 400):   # a quarter of a second of back-to-back launches
                for _ in range(5):
                    db.fill(ctx, h, kernel, order_after_current=False)
                n_alone += 5
            torch.cuda.synchronize()
            dt = time.perf_counter() - t_a
            scale_base = {"workload": f"{workload} share of one GPU ({batch.n_pairs} pairs), rank 0 alone, {n_alone} steps in {dt:.2f} s",
                          "per_gpu_alone_gcups": batch.cells() * n_alone / dt / 1e9}
        grp.barrier()

    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        starts[i].record(db.stream)  # the stream the kernel is launched on
        db.fill(ctx, h, kernel, order_after_current=False)
        ends[i].record(db.stream)
    torch.cuda.synchronize()
    grp.barrier()
    elapsed = time.perf_counter() - t0

    elapsed = grp.max_float(elapsed)               # MAX over ranks
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in zip(starts, ends)]))
    kern_ms_max = grp.max_float(kern_ms)

    # parity spot check outside the timed region (checker = oracle): bit-exact int32
    import orclib as O
    osc = O.Scoring.from_buffer_copy(bytes(sc))
    bit_exact = True
    for p in range(0, batch.n_pairs, max(1, batch.n_pairs // 16)):
        rc, M, A, B = O.oracle_fill(osc, batch.seq_a(p), batch.seq_b(p), is_sw)
        gM, gA, gB = db.pair_matrices(p)
        bit_exact &= rc == 0 and np.array_equal(M, gM) and np.array_equal(A, gA) and np.array_equal(B, gB)
    bit_exact = grp.sum_int(0 if bit_exact else 1) == 0

    # What the SAME kernel does on arenas nobody placed: three plain arenas (option arena_scan_gib = 0: no walk), a few steps --
    # so that the line states what the placement contributes instead of implying it (roofline.frac_unplaced).  Measured LAST:
    # an allocation of gigabytes before the walk changes what the allocator hands the walk (one box of four ended at quality
    # 1.012 / 0.815 with it in front).  The arenas come from the chunks the walk left with the process when there are any --
    # consecutive 512 MiB chunks of the allocation order, which is what three hipMallocs in a row are too.
    unplaced = None
    if args.placement == "spread" and not args.no_unplaced and world == 1:
        with ctx.options(arena_scan_gib=0):
            dbp = S.DeviceBatch(batch, local, placement="spread", ctx=ctx)
        ms = dbp.time_fill_ms(ctx, h, kernel, 6)[2:]
        unplaced = {"kernel_ms": float(np.mean(ms)), "steps": len(ms), "kernel": S.KERNEL_NAMES[kernel],
                    "arenas": "three plain arenas, no walk (seqalign_arenas_alloc with arena_scan_gib = 0), measured after the timed loop"}
        del dbp
        torch.cuda.synchronize()

    # per-rank figures for the report: a slow rank (placement, NUMA, a busy neighbour) must be visible, not averaged away
    per_rank = grp.gather_objects({"rank": rank, "device": local, "kernel_ms": round(kern_ms, 4),
                                   "arena_placement_quality": round(db.placement_quality, 3),
                                   "arena_placement_tries": (db.placement_info or {}).get("tries"),
                                   "e2e_ms": round(e2e["ms_this_rank"], 3) if e2e else None, **pin})
    out = None
    if rank == 0:
        alg_bytes = db.algorithmic_bytes()
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        prof = ROOT / "profiles" / "pmc_traffic.json"
        if prof.exists():
            try:
                t = json.loads(prof.read_text())
                key = f"{workload}:{S.KERNEL_NAMES[kernel]}:{batch.n_pairs}"
                if key in t:
                    traffic = t[key].get("hbm_bytes_per_launch")
                    traffic_source = (f"static: profiles/pmc_traffic.json[{key}] <- {t[key].get('source', 'rocprofv3 --pmc passes')}"
                                      " (PMC counters cannot be read from inside this process)")
            except Exception:
                traffic = None
        out = {
            "metric": "dp_cell_updates_per_sec", "value": total_cells * steps / elapsed / 1e9, "unit": "GCUPS",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{workload}: {desc}", "pairs_per_gpu": batch.n_pairs,
                       "global_pairs": global_pairs, "kernel": S.KERNEL_NAMES[kernel],
                       "parallelism": f"pair-sharded x{world}, no data-path collective; control plane: {backend}",
                       "arena_placement": args.placement, "arena_placement_quality": round(db.placement_quality, 3),
                       "arena_placement_search": db.placement_info,
                       "cells_per_step_per_gpu": batch.cells(), "ranks_share_devices": folded,
                       **({"scale_base": scale_base} if scale_base else {})},
            "bit_exact_vs_oracle": bool(bit_exact),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": kern_ms, "kernel_ms_slowest_rank": kern_ms_max,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if unplaced:
            out["roofline"]["frac_unplaced"] = alg_bytes / (unplaced["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["unplaced"] = unplaced
        if e2e:
            e2e.pop("ms_this_rank", None)
            rec = ROOT / "profiles" / "e2e_roofline.json"
            if rec.exists():
                try:
                    table = json.loads(rec.read_text())
                    r = table.get(f"{workload}:{batch.n_pairs}")
                    #  these come from a committed rocprofv3 record, not from this run: say so in the line
                    tag = {"measured_in_this_run": False,
                           "pre_recorded": f"profiles/e2e_roofline.json (rocprofv3 kernel trace + PMC passes, {table.get('_recorded', 'round unknown')})"}
                    if r:
                        e2e["roofline"] = {**r, **tag}     # bound, kernel, kernel_ms, instructions, frac + where they come from
                    r4 = table.get(f"{workload}:{batch.n_pairs}:hits4")
                    if r4 and "up_to_4_hits" in e2e:
                        e2e["up_to_4_hits"]["roofline"] = {**r4, **tag}
                except Exception:
                    pass
            out["e2e"] = e2e
        out["per_rank"] = per_rank
        worst_q = min((r["arena_placement_quality"] for r in per_rank if r["arena_placement_quality"] is not None
                       and r["arena_placement_quality"] >= 0), default=None)
        if worst_q is not None and db.placement_info and worst_q < 0.97:
            out["config"]["arena_placement_note"] = (f"placement quality below 0.97 on at least one rank (worst {worst_q}): the arenas "
                                                     "disturb each other and the walk found nothing better; see per_rank / "
                                                     "arena_placement_search")
        if is_sw:
            # the SW local-maxima reduction as a separate kernel (4 B/cell read)
            thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
            for i in range(11):
                ev[i].record(db.stream)
                db.sw_reduce_launch(ctx, thr)
            ev[11].record(db.stream)
            torch.cuda.synchronize()
            rms = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(1, 11)]))
            rbytes = int(4 * db.cells_host.sum())
            out["sw_reduce"] = {"min_score": thr, "kernel_ms": rms, "bound": "hbm",
                                "achieved": rbytes / (rms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": rbytes / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "algorithmic_bytes_per_launch": rbytes}
        out["_batch"] = batch     # (for the caller's cpu_baseline leg; popped before the line is printed)

    grp.barrier()
    ctx.release_scoring(h)
    del db
    torch.cuda.synchronize()
    ctx.close()
    return out if rank == 0 else None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)   # C2: ~0.42 s of kernels; the whole default run stays under a minute
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=list(WORKLOADS),
                    help="default: C2 at --gpus 1, C5 (125k pairs per GPU of the 1M-pair batch) at --gpus N > 1")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="C5 at N > 1: weak = 125k pairs per GPU (default), strong = 1M pairs over the N ranks")
    ap.add_argument("--kernel", default="auto", choices=["auto", "wavefront", "rowscan", "stream", "strips", "wgstream"])
    ap.add_argument("--all-kernels", action="store_true",
                    help="--kernel auto also times wavefront / rowscan / strips (correctness paths that have never won a configuration)")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="N = 1, default workload: skip the `configs` block (C3, C4, C5's share measured beside the C2 headline)")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip e2e.stream (the host-level call as a stream of submitted batches)")
    ap.add_argument("--stream-batches", type=int, default=32)
    ap.add_argument("--no-unplaced", action="store_true",
                    help="skip the few steps of the same kernel on unplaced arenas (roofline.frac_unplaced)")
    ap.add_argument("--e2e-after", type=float, default=2.0,
                    help="seconds after the arena placement before `e2e` is measured (the driver's transient after the placement walk)")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not pin the rank to its GPU's NUMA node")
    ap.add_argument("--placement", default="spread", choices=["spread", "packed"],
                    help="output arenas: the library's spread allocator (seqalign_arenas_alloc) or one packed allocation")
    ap.add_argument("--plumbing-test", action="store_true",
                    help="CPU-only check of launcher + sharding + reductions (tests/test_multigpu_cpu.py); not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])      # no launcher around us: start the ranks ourselves
    return run(args)


if __name__ == "__main__":
    sys.exit(main())
