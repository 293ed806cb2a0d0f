#!/usr/bin/env python3
"""Fill throughput on few, long pairs (the regime one wave per pair is worst at)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402

ctx = S.Context(0)
sc = S.make_scoring({"preset": "default"})
h = ctx.upload_scoring(sc, 0)
sizes = [(1, 10000), (1, 20000), (16, 5000), (200, 2000), (512, 2000), (2000, 2000), (1000, 1000), (256, 1000), (64, 1000), (64, 600), (512, 600), (4096, 600)]
if len(sys.argv) > 1 and sys.argv[1] == 'mid':   # 513..1023 columns: stream (12/16 columns per lane) vs wgstream
    sizes = [(64, 1000), (256, 1000), (1000, 1000), (4000, 1000), (512, 600), (4096, 600), (20000, 600), (2000, 800)]
if len(sys.argv) > 1 and sys.argv[1] == 'wg':   # the regime of sa_fill_wgstream.hip
    sizes = [(16, 2000), (64, 2000), (200, 2000), (512, 2000), (2000, 2000), (100, 4000), (500, 4000), (64, 1500), (1000, 1500), (4000, 1200)]
for n, length in sizes:
    batch = W.dna_nw_150(n, seed=9, length=length)
    db = S.DeviceBatch(batch, 0, ctx=ctx)
    reps = 3 if n * length * length > 5e7 else 10
    db.time_fill_ms(ctx, h, S.KERNEL_AUTO, 1)
    ms = float(np.median(db.time_fill_ms(ctx, h, S.KERNEL_AUTO, reps)))
    alg = db.algorithmic_bytes()
    print(f"{n:5d} x {length}x{length}: {ms:9.3f} ms  {batch.cells() / ms / 1e6:8.2f} GCUPS  {alg / ms / 1e6:7.0f} GB/s", flush=True)
    del db
