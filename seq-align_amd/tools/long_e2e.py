#!/usr/bin/env python3
"""seqalign_nw_batch end to end (fill + device traceback + strings back) on one long pair and on C2."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402

ctx = S.Context(0)
sc = S.make_scoring({"preset": "default"})
for n, length in [(1, 10000), (16, 5000), (10000, 150), (100000, 150)]:
    batch = W.dna_nw_150(n, seed=9, length=length, related=True)
    ctx.nw_batch(batch, sc, raw=True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ctx.nw_batch(batch, sc, raw=True); ts.append(time.perf_counter() - t0)
    print(f"nw_batch {n:6d} x {length}x{length}: {min(ts) * 1e3:9.3f} ms  {batch.cells() / min(ts) / 1e9:8.2f} GCUPS", flush=True)
