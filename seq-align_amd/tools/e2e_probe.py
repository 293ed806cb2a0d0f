#!/usr/bin/env python3
"""What changes seqalign_nw_batch's wall clock around it, one experiment per run (the records: profiles/r03/r03_stream_queue_sharing.txt,
r03_after_placement_transient.txt).  Two families of experiments, first argument picks one:

  steps  <variant> [pairs] [seconds]   bench.py's steps before its `e2e` measurement, one variant per run: which of them costs the call
         its speed.  variant = full | no_choice | no_arenas | packed | only_wavefront | only_rowscan | only_strips | only_wgstream |
         only_stream | only_stream_ctxstream
  phases <mode>                        one process, phase by phase: fresh context, torch.cuda initialised, arenas placed, streams
         created.  mode = arenas_first | stream_first | stream | arenas_plain | arenas_vmm (the transient after the placement
         walk, timed every 0.3 s for 8 s) | tensors
(Rounds 1-3 kept these as e2e_probe.py, e2e_probe2.py and e2e_probe3.py.)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np, torch
import seqalign_amd as S
from seqalign_amd import workloads as W

family = sys.argv[1] if len(sys.argv) > 1 else ""
if family not in ("steps", "phases") or len(sys.argv) < 3:
    raise SystemExit(__doc__)
sys.argv = [sys.argv[0]] + sys.argv[2:]


def steps():
    variant = sys.argv[1]
    torch.cuda.set_device(0)
    n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 125000
    batch = W.dna_nw_indexed(0, n_pairs, seed=5, length=150)
    ctx = S.Context(0)
    sc = S.make_scoring({"preset": "default"})
    h = ctx.upload_scoring(sc, 0)
    db = None
    if variant != "no_arenas":
        db = S.DeviceBatch(batch, 0, placement="packed" if variant == "packed" else "spread", ctx=ctx)
        names = {"only_wavefront": [S.KERNEL_WAVEFRONT], "only_rowscan": [S.KERNEL_ROWSCAN], "only_stream": [S.KERNEL_STREAM],
                 "only_strips": [S.KERNEL_STRIPS], "only_wgstream": [S.KERNEL_WGSTREAM], "no_choice": [], "packed": [],
                 "only_stream_ctxstream": [S.KERNEL_STREAM]}
        for k in names.get(variant, [S.KERNEL_WAVEFRONT, S.KERNEL_ROWSCAN, S.KERNEL_STREAM, S.KERNEL_STRIPS, S.KERNEL_WGSTREAM]):
            if variant.endswith("_ctxstream"):      # the same launches on the context's own stream instead of a torch stream
                import ctypes as C
                ms = (C.c_float * 6)()
                S._check(S.lib().seqalign_time_fill_ms(ctx._h, h, C.byref(db.desc), C.c_int(k), C.c_void_p(0), C.c_int(6), ms), "x")
            else:
                db.time_fill_ms(ctx, h, k, 6)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < (float(sys.argv[3]) if len(sys.argv) > 3 else 7):
        ctx.nw_batch(batch, sc, raw=True)
    ts = []
    for it in range(8):
        t1 = time.perf_counter(); ctx.nw_batch(batch, sc, raw=True); ts.append((time.perf_counter() - t1) * 1e3)
    print(variant, " ".join("%.3f" % t for t in ts), flush=True)


def phases():
    batch = W.dna_nw_indexed(0, 125000, seed=5, length=150)
    sc = S.make_scoring({"preset": "default"})

    def loop(ctx, tag):
        ts = []
        for it in range(8):
            t0 = time.perf_counter(); ctx.nw_batch(batch, sc, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
        print(tag, " ".join("%.2f" % t for t in ts[3:]), flush=True)
    ctx = S.Context(0)
    mode = sys.argv[1]
    if mode == "arenas_first":     # what bench.py does: the placed arenas BEFORE the first host-level call allocates its buffers
        import ctypes as C
        ptrs = (C.c_void_p * 3)(); q = C.c_float(-1.0)
        S._check(S.lib().seqalign_arenas_alloc(ctx._h, C.c_uint64(4 * 2850125000 // 1024 * 1024), ptrs, C.byref(q)), "x")
        time.sleep(7)
        loop(ctx, "first calls 7 s after the placed arenas")
        loop(ctx, "again                                  ")
        return
    if mode == "stream_first":     # torch's stream pool (32 + 32 HIP streams) exists before the library creates its own streams
        torch.cuda.set_device(0)
        s = torch.cuda.Stream(torch.device("cuda", 0))
        loop(ctx, "first calls after torch.cuda.Stream()")
        loop(ctx, "again                                ")
        return
    loop(ctx, "fresh context                 ")
    torch.cuda.set_device(0)
    x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    loop(ctx, "after torch.cuda init         ")
    if mode == "stream":
        s = torch.cuda.Stream(torch.device("cuda", 0))
        loop(ctx, "after torch.cuda.Stream()     ")
    elif mode == "arenas_plain":
        ctx.set_option("arena_scan_gib", 0)
        import ctypes as C
        ptrs = (C.c_void_p * 3)(); q = C.c_float(-1.0)
        S._check(S.lib().seqalign_arenas_alloc(ctx._h, C.c_uint64(4 * 2850125000 // 1024 * 1024), ptrs, C.byref(q)), "x")
        loop(ctx, "after plain arenas (no VMM)   ")
    elif mode == "arenas_vmm":
        import ctypes as C
        ptrs = (C.c_void_p * 3)(); q = C.c_float(-1.0)
        S._check(S.lib().seqalign_arenas_alloc(ctx._h, C.c_uint64(4 * 2850125000 // 1024 * 1024), ptrs, C.byref(q)), "x")
        t_begin = time.perf_counter()
        while time.perf_counter() - t_begin < 8:
            loop(ctx, "%.1f s after the placed arenas (VMM walk)" % (time.perf_counter() - t_begin))
            time.sleep(0.3)
    elif mode == "tensors":
        t = [torch.from_numpy(batch.arena).to("cuda"), torch.zeros(125000, dtype=torch.int64, device="cuda")]
        loop(ctx, "after torch tensors           ")


steps() if family == "steps" else phases()
