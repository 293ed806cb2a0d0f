#!/usr/bin/env python3
"""sw_reduce_kernel (the SW local-maxima reduction as its own pass over match_scores, 4 B per cell read) on C3 / C4 against what
this GPU reads at all: the same bytes through torch's sum (a library reduction) and through a plain grid-stride max of the whole
arena.  Prints ms and TB/s per variant."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402
from bench import WORKLOADS  # noqa: E402


def timed(fn, stream, reps=12):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    with torch.cuda.stream(stream):
        for i in range(reps):
            ev[i].record(stream)
            fn()
        ev[reps].record(stream)
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(2, reps))
    return ts[len(ts) // 2]


with S.Context(0) as ctx:
    for name in sys.argv[1:] or ["C3", "C4"]:
        gen, kwargs, n, is_sw, spec, _ = WORKLOADS[name]
        batch = getattr(W, gen)(n, **kwargs)
        sc = S.make_scoring(spec)
        h = ctx.upload_scoring(sc, 1)
        db = S.DeviceBatch(batch, 0, ctx=ctx)
        db.fill(ctx, h, S.KERNEL_STREAM)
        torch.cuda.synchronize()
        thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
        nbytes = int(4 * db.cells_host.sum())
        M = db.M[: nbytes // 4]
        rows = []
        for depth in (0, 4, 8):
            ctx.set_option("reduce_depth", depth)
            rows.append((f"sw_reduce depth {depth}", timed(lambda: db.sw_reduce_launch(ctx, thr), db.stream)))
        ctx.set_option("reduce_depth", 0)
        rows.append(("torch max", timed(lambda: M.max(), db.stream)))
        for what, t in rows:
            print(f"{name} {what:18s} {t:7.4f} ms  {nbytes / t / 1e9:7.3f} TB/s  {nbytes / t / 1e9 / 8:6.3f} of 8 TB/s", flush=True)
        ctx.release_scoring(h)
        del db
