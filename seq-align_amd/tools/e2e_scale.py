#!/usr/bin/env python3
"""End-to-end nw_batch throughput vs batch size (host buffers in, strings out)."""
import json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import seqalign_amd as S
from seqalign_amd import workloads as W
out = {}
with S.Context(0) as ctx:
    sc = S.make_scoring({"preset": "default"})
    for n in (1000, 10000, 125000, 500000):
        if n > 125000: ctx.nw_batch(W.dna_nw_150(n, seed=2), sc, raw=True)   # warm the buffers
        batch = W.dna_nw_150(n, seed=1)
        ctx.nw_batch(batch, sc, raw=True) if n <= 125000 else None
        dts = []
        for _ in range(3):   # min of 3: the first calls after the arenas were (re)placed are slower
            t0 = time.perf_counter(); ctx.nw_batch(batch, sc, raw=True); dts.append(time.perf_counter() - t0)
        dt = min(dts)
        out[n] = dict(ms=dt * 1e3, gcups=batch.cells() / dt / 1e9, pairs_per_s=n / dt)
        print(n, out[n], flush=True)
