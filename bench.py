#!/usr/bin/env python3
"""bench.py -- GCUPS of the DP-fill hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (seqalign_fill_batch_device: the fill of the
match / gap_a / gap_b matrices, reference src/alignment.c:28-168) over one
synthetic batch that is already resident in HBM.  Workload at N=1: BASELINE
configs[1] -- 10 000 NW pairs, DNA 150x150, default scoring 1/-2/-4/-1.
For N>1 (launched by torch.distributed.run, one rank per GPU) every rank fills
its own 10 000-pair shard of a 10 000*N-pair batch: weak scaling, independent
pairs, NO collective on the data path (the only collectives are the barrier and
the MAX over ranks of the elapsed time).

    python bench.py --gpus 1 --steps 200 --warmup 10

Prints ONE JSON line (rank 0).  `value` = all ranks' cells / max-rank seconds.
`roofline`  : algorithmic bytes per launch / mean kernel duration (HIP events on
              the launch stream, recorded inside the timed region) vs 8 TB/s.
`cpu_baseline`: the reference itself (oracle/_ref, built from /root/reference in
              the authoring container) or, if absent, our C restatement
              (oracle/), timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for p in (ROOT / "seq-align_amd" / "python", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import seqalign_amd as S                     # noqa: E402
from seqalign_amd import workloads as W      # noqa: E402

HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec

WORKLOADS = {
    # name: (generator, kwargs, pairs per GPU, is_sw, scoring spec)
    "C2": ("dna_nw_150", dict(seed=1), 10000, 0, {"preset": "default"},
           "10k NW pairs, DNA 150x150, default scoring 1/-2/-4/-1 (BASELINE configs[1])"),
    "C3": ("dna_sw_read_vs_ref", dict(seed=2), 10000, 1, {"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]},
           "10k SW pairs, DNA 150x1000 read-vs-ref, 2/-2/-2/-1 (BASELINE configs[2])"),
    "L1000": ("dna_nw_150", dict(seed=4, length=1000), 1000, 0, {"preset": "default"},
              "1k NW pairs, DNA 1000x1000 (long-sequence check, not a BASELINE config)"),
    "C4": ("protein_sw_300", dict(seed=3), 4000, 1, {"preset": "BLOSUM62"},
           "4k SW pairs, protein 300x300, BLOSUM62 (BASELINE configs[3])"),
    "C5": ("dna_nw_150", dict(seed=5), 125000, 0, {"preset": "default"},
           "1M NW pairs, DNA 150x150, sharded over 8 GPUs: 125k pairs (34 GB of matrices) per GPU (BASELINE configs[4])"),
}


def _ref_worker(args):
    """aligner_align of the compiled reference over a slice of pairs, own aligner_t
    (the reference is re-entrant per aligner object, SURVEY 8b threading).  ctypes
    releases the GIL for the duration of each call, so threads run in parallel."""
    import orclib as O
    ref, sc, pairs, isw, deadline = args
    al = O.Aligner()
    C.memset(C.byref(al), 0, C.sizeof(al))
    cells = done = 0
    while time.perf_counter() < deadline:
        for a, b in pairs:
            ref.aligner_align(C.byref(al), a, b, C.c_size_t(len(a)), C.c_size_t(len(b)), C.byref(sc), isw)
            cells += len(a) * len(b)
            done += 1
    ref.aligner_destroy(C.byref(al))
    return cells, done


def cpu_baseline(batch, spec, is_sw, budget_s=20.0):
    """Reference CPU path on THIS host, bounded sample.  checker code: allowed here."""
    import orclib as O
    n = batch.n_pairs
    ref = O.ref()
    if ref is not None:
        sc = O.build_scoring(spec, "ref")
        bufs = [(batch.seq_a(p), batch.seq_b(p)) for p in range(n)]
        isw = C.c_char(bytes([is_sw]))
        t0 = time.perf_counter()
        cells, done = _ref_worker((ref, sc, bufs, isw, t0 + budget_s * 0.4))
        dt = time.perf_counter() - t0
        out = dict(value=cells / dt / 1e9, unit="GCUPS", cores=1, kind="reference",
                   sample=f"{done} pairs ({cells} cells) through aligner_align of the compiled reference "
                          f"(oracle/_ref), 1 thread, {dt:.1f} s")
        # the same on every host core (one aligner_t per thread), reported next to the 1-thread figure
        from concurrent.futures import ThreadPoolExecutor
        cores = os.cpu_count() or 1
        per = max(1, n // cores)
        slices = [bufs[i * per:(i + 1) * per] or bufs[:per] for i in range(cores)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            res = list(ex.map(_ref_worker, [(ref, sc, sl, isw, t0 + budget_s * 0.25) for sl in slices]))
        dt = time.perf_counter() - t0
        out["all_cores"] = dict(value=sum(r[0] for r in res) / dt / 1e9, unit="GCUPS", cores=cores,
                                sample=f"{sum(r[1] for r in res)} pairs over {cores} threads, {dt:.1f} s")
        return out
    import seqalign_amd
    sc = O.Scoring.from_buffer_copy(bytes(seqalign_amd.make_scoring(spec)))
    cells, secs, reps = 0, 0.0, 0
    while secs < budget_s * 0.5:
        chk = C.c_uint64(0)
        secs += O.oracle().orc_time_fill_batch(C.byref(sc), batch.arena.ctypes.data_as(C.c_char_p),
                                               batch.off_a.ctypes.data_as(C.c_void_p), batch.len_a.ctypes.data_as(C.c_void_p),
                                               batch.off_b.ctypes.data_as(C.c_void_p), batch.len_b.ctypes.data_as(C.c_void_p),
                                               C.c_size_t(n), C.c_int(is_sw), C.byref(chk))
        cells += batch.cells()
        reps += 1
    return dict(value=cells / secs / 1e9, unit="GCUPS", cores=1, kind="port",
                sample=f"{reps}x{n} pairs ({cells} cells) through orc_fill (oracle/, C restatement), 1 thread, {secs:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C2", choices=list(WORKLOADS))
    ap.add_argument("--kernel", default="auto", choices=["auto", "wavefront", "rowscan", "stream", "strips", "wgstream"])
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--placement", default="spread", choices=["spread", "packed"],
                    help="output arenas: the library's spread allocator (seqalign_arenas_alloc) or one packed allocation")
    args = ap.parse_args()

    import torch
    from seqalign_amd.dist import Group, env_world

    rank, local, world = env_world()
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with python -m torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the product has no CPU path")
    # SEQALIGN_DIST_BACKEND=gloo (+ ranks folded onto the visible GPUs) exists only to
    # exercise the multi-process path on a 1-GPU box; the driver's runs use RCCL.
    backend = os.environ.get("SEQALIGN_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    grp = Group(backend, torch.device("cuda", local) if backend == "nccl" else None)   # "nccl" is RCCL on ROCm

    gen, kwargs, per_gpu, is_sw, spec, desc = WORKLOADS[args.workload]
    per_gpu = args.pairs or per_gpu
    # one global batch, sharded by contiguous pair index: rank g owns [g*n/G, (g+1)*n/G)
    batch = getattr(W, gen)(per_gpu * world, **kwargs).shard(rank, world)

    lib = S.lib()
    ctx = S.Context(local)
    sc = S.make_scoring(spec)
    h = ctx.upload_scoring(sc, is_sw)
    db = S.DeviceBatch(batch, local, placement=args.placement, ctx=ctx)   # arenas from seqalign_arenas_alloc

    # kernel choice: measured, not guessed
    if args.kernel == "auto":
        best = None
        for k in (S.KERNEL_WAVEFRONT, S.KERNEL_ROWSCAN, S.KERNEL_STREAM, S.KERNEL_STRIPS, S.KERNEL_WGSTREAM):
            ms = db.time_fill_ms(ctx, h, k, 6)[1:]
            m = float(np.median(ms))
            if best is None or m < best[1]:
                best = (k, m)
        kernel = grp.broadcast_int(best[0], 0)   # all ranks run the same kernel
    else:
        kernel = {"wavefront": S.KERNEL_WAVEFRONT, "rowscan": S.KERNEL_ROWSCAN, "stream": S.KERNEL_STREAM,
                  "strips": S.KERNEL_STRIPS, "wgstream": S.KERNEL_WGSTREAM}[args.kernel]

    barrier = grp.barrier

    torch.cuda.synchronize()
    for _ in range(args.warmup):
        db.fill(ctx, h, kernel, order_after_current=False)
    torch.cuda.synchronize()
    barrier()

    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record(db.stream)  # the stream the kernel is launched on
        db.fill(ctx, h, kernel, order_after_current=False)
        ends[i].record(db.stream)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    elapsed = grp.max_float(elapsed)               # MAX over ranks
    total_cells = grp.sum_int(batch.cells())       # whole-job cells per step

    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in zip(starts, ends)]))

    # parity spot check outside the timed region (checker = oracle): bit-exact int32
    bit_exact = None
    if rank == 0:
        import orclib as O
        osc = O.Scoring.from_buffer_copy(bytes(sc))
        bit_exact = True
        for p in range(0, batch.n_pairs, max(1, batch.n_pairs // 16)):
            rc, M, A, B = O.oracle_fill(osc, batch.seq_a(p), batch.seq_b(p), is_sw)
            gM, gA, gB = db.pair_matrices(p)
            bit_exact &= rc == 0 and np.array_equal(M, gM) and np.array_equal(A, gA) and np.array_equal(B, gB)

    if rank == 0:
        alg_bytes = db.algorithmic_bytes()
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        prof = ROOT / "profiles" / "pmc_traffic.json"
        if prof.exists():
            try:
                t = json.loads(prof.read_text())
                key = f"{args.workload}:{S.KERNEL_NAMES[kernel]}:{batch.n_pairs}"
                traffic = t.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "dp_cell_updates_per_sec", "value": total_cells * args.steps / elapsed / 1e9, "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "pairs_per_gpu": batch.n_pairs,
                       "global_pairs": batch.n_pairs * world, "kernel": S.KERNEL_NAMES[kernel],
                       "parallelism": f"pair-sharded x{world}, no collective",
                       "arena_placement": args.placement, "arena_placement_quality": round(db.placement_quality, 3),
                       "cells_per_step_per_gpu": batch.cells()},
            "bit_exact_vs_oracle": bool(bit_exact),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if is_sw:
            # the SW local-maxima reduction, a separate kernel (4 B/cell read)
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
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch, spec, is_sw)
        print(json.dumps(out), flush=True)

    ctx.release_scoring(h)
    ctx.close()
    grp.close()


if __name__ == "__main__":
    main()
