#!/usr/bin/env python3
"""Kernel time as a function of the number of pairs (same pair shape): separates
the per-pair cost from the fixed ramp/drain cost of one launch, and shows the
single-pair latency (1 wave per CU).

    python seq-align_amd/tools/size_sweep.py --workload C2 --kernel stream 256 1024 2048 ..."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402
from bench import WORKLOADS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("sizes", nargs="+", type=int)
ap.add_argument("--workload", default="C2")
ap.add_argument("--kernel", default="stream")
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--budget-ms", type=float, default=150.0)
args = ap.parse_args()
gen, kwargs, per_gpu, is_sw, spec, _ = WORKLOADS[args.workload]
KID = {"wavefront": S.KERNEL_WAVEFRONT, "rowscan": S.KERNEL_ROWSCAN, "stream": S.KERNEL_STREAM}[args.kernel]
ctx = S.Context(0)
h = ctx.upload_scoring(S.make_scoring(spec), is_sw)
rows = []
for n in args.sizes:
    batch = getattr(W, gen)(n, **kwargs)
    db = S.DeviceBatch(batch, 0)
    reps = max(args.launches, int(args.budget_ms / max(0.05, 0.5 * n / 10000)))   # ~budget_ms of GPU time per size
    db.time_fill_ms(ctx, h, KID, reps // 2)                                       # warm-up: clocks settle
    xs = db.time_fill_ms(ctx, h, KID, reps)
    ms = float(np.median(xs))
    alg = db.algorithmic_bytes()
    rows.append(dict(pairs=n, ms=ms, gcups=batch.cells() / ms / 1e6, tbps=alg / ms / 1e9))
    print(f"{n:8d} pairs  x{reps:5d}  min {min(xs):8.4f}  {ms:8.4f} ms  {rows[-1]['gcups']:7.1f} GCUPS  {rows[-1]['tbps']:6.3f} TB/s", flush=True)
    del db
print(json.dumps(rows))
