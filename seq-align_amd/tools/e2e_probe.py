#!/usr/bin/env python3
"""seqalign_nw_batch (C5's share) in one process, phase by phase of what bench.py does before it measures `e2e`: fresh
context, torch.cuda initialised, the placed arenas allocated, fills run, the kernel choice -- which step changes the call."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np, torch
import seqalign_amd as S
from seqalign_amd import workloads as W
batch = W.dna_nw_indexed(0, 125000, seed=5, length=150)
sc = S.make_scoring({"preset": "default"})
def loop(ctx, tag):
    ts = []
    for it in range(8):
        t0 = time.perf_counter(); ctx.nw_batch(batch, sc, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, " ".join("%.2f" % t for t in ts[3:]), flush=True)

ctx = S.Context(0)
loop(ctx, "fresh context              ")
torch.cuda.set_device(0)
x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
loop(ctx, "after torch.cuda init      ")
db = S.DeviceBatch(batch, 0, placement="spread", ctx=ctx)
loop(ctx, "after DeviceBatch (arenas) ")
t_w = time.perf_counter()
while time.perf_counter() - t_w < 7:
    ctx.nw_batch(batch, sc, raw=True)
loop(ctx, "... 7 s later              ")
h = ctx.upload_scoring(sc, False)
for _ in range(30):
    db.fill(ctx, h, S.KERNEL_STREAM, order_after_current=False)
torch.cuda.synchronize()
loop(ctx, "after 30 three-matrix fills")
for k in (S.KERNEL_WAVEFRONT, S.KERNEL_ROWSCAN, S.KERNEL_STREAM, S.KERNEL_STRIPS, S.KERNEL_WGSTREAM):
    ms = db.time_fill_ms(ctx, h, k, 6)
    loop(ctx, "after time_fill_ms kernel %d (%.2f ms)" % (k, float(np.median(ms[1:]))))
t_end = time.perf_counter() + 0.25
while time.perf_counter() < t_end:
    ctx.nw_batch(batch, sc, raw=True)
loop(ctx, "after 0.25 s of calls      ")
del db
loop(ctx, "after freeing the arenas   ")
