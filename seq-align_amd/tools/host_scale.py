#!/usr/bin/env python3
"""The HOST side of N = 8: eight processes, pinned as bench.py's pin_rank would pin eight ranks, each running only the host
legs of seqalign_nw_batch on a 125 000-pair share of BASELINE config 5 (sizes + offsets + packing; expansion of the moves
into strings) -- no GPU involved (seqalign_host_legs_nw).  Reports the legs' time per call for one process alone and for
N at once: a rank must sustain its share's pack + expand well inside the ~3.5 ms the GPU needs for the share.
    python host_scale.py [--ranks 8] [--pairs 125000] [--iters 20]
Runs on any box (the CPU container, or the GPU box's host)."""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))


def worker(rank, n_ranks, pairs, iters, cpus, threads, start, q):
    if cpus:
        os.sched_setaffinity(0, cpus)          # before the library creates its pool: the workers inherit the mask
    if threads and "SEQALIGN_HOST_THREADS" not in os.environ:
        os.environ["SEQALIGN_HOST_THREADS"] = str(threads)   # bench.py: a rank's share of the container's CPU quota
    import numpy as np
    import seqalign_amd as S
    from seqalign_amd import workloads as W
    batch = W.dna_nw_indexed(rank * pairs, pairs, seed=5)
    caps = batch.len_a.astype(np.uint64) + batch.len_b.astype(np.uint64) + np.uint64(1)
    str_off = np.zeros(pairs, np.uint64)
    str_off[1:] = np.cumsum(caps)[:-1]
    total = int(caps.sum()) + 1
    out_a, out_b = np.zeros(total, np.uint8), np.zeros(total, np.uint8)
    out_len = np.zeros(pairs, np.uint32)
    d = S.batch_desc(batch)
    pack, expand = C.c_double(0), C.c_double(0)
    lib = S.lib()
    call = lambda it: lib.seqalign_host_legs_nw(C.byref(d), S._ptr(str_off), S._ptr(out_a), S._ptr(out_b), S._ptr(out_len),
                                                C.c_int(it), C.byref(pack), C.byref(expand))
    assert call(3) == 0                       # warm up: pool threads, page faults of the outputs
    start.wait()
    t0 = time.perf_counter()
    assert call(iters) == 0
    q.put({"rank": rank, "cpus": len(cpus) if cpus else None, "pack_ms": pack.value, "expand_ms": expand.value,
           "wall_ms_per_call": (time.perf_counter() - t0) * 1e3 / iters, "simd_expand": bool(lib.sa_moves_uses_simd())})


def run(n_ranks, pairs, iters, shares, threads):
    ctx = mp.get_context("spawn")
    start, q = ctx.Event(), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, n_ranks, pairs, iters, shares[r], threads, start, q)) for r in range(n_ranks)]
    for p in ps:
        p.start()
    time.sleep(8 if pairs > 50000 else 3)     # everybody has generated its share and warmed up
    start.set()
    out = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join()
    return sorted(out, key=lambda r: r["rank"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=125000)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    allowed = sorted(os.sched_getaffinity(0))
    # bench.py pin_rank: the CPUs of a GPU's NUMA node dealt out evenly among the ranks on it; without GPUs here, deal the
    # allowed CPUs out evenly in contiguous runs (hyper-thread siblings are c and c + n/2 on the GPU box: same slice of both halves)
    half = len(allowed) // 2
    shares = []
    for r in range(a.ranks):
        if len(allowed) >= 2 * a.ranks and len(allowed) % 2 == 0:
            per = half // a.ranks
            shares.append(allowed[r * per:(r + 1) * per] + allowed[half + r * per:half + (r + 1) * per])
        else:
            per = max(1, len(allowed) // a.ranks)
            shares.append(allowed[(r * per) % len(allowed):(r * per) % len(allowed) + per] or allowed)
    from bench import cgroup_cpu_quota
    quota = cgroup_cpu_quota()
    threads = min(32, len(shares[0]), max(2, quota // a.ranks)) if quota else None   # what bench.py gives a rank at N = ranks
    alone = run(1, a.pairs, a.iters, [shares[0]], threads)
    together = run(a.ranks, a.pairs, a.iters, shares, threads)
    legs = lambda r: r["pack_ms"] + r["expand_ms"]
    print(json.dumps({"host_cpus": len(allowed), "ranks": a.ranks, "pairs_per_rank": a.pairs, "iterations": a.iters,
                      "cpus_per_rank": len(shares[0]), "cgroup_cpu_quota": quota, "host_threads_per_rank": threads,
                      "alone": alone[0], "together": together,
                      "legs_ms_alone": round(legs(alone[0]), 3),
                      "legs_ms_together_worst": round(max(legs(r) for r in together), 3),
                      "pairs_per_s_per_rank_alone": round(a.pairs / legs(alone[0]) * 1e3),
                      "pairs_per_s_per_rank_together_worst": round(a.pairs / max(legs(r) for r in together) * 1e3)}))


if __name__ == "__main__":
    main()
