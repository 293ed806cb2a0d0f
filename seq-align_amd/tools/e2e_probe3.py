#!/usr/bin/env python3
"""bench.py's steps before its `e2e` measurement, one variant per run: which of them costs seqalign_nw_batch its speed.
    python e2e_probe3.py <variant>     full | no_choice | no_arenas | packed | only_wavefront | only_rowscan | only_strips | only_wgstream | only_stream"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np, torch
import seqalign_amd as S
from seqalign_amd import workloads as W
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
