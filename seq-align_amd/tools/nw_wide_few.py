#!/usr/bin/env python3
"""FEW wide NW pairs: the direction-byte path (nw_dirs = 1, one wave per pair) against three matrices (nw_dirs = 0: several waves per
pair), alternating in one process.      nw_wide_few.py [len = 1000] [pairs ...]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np
import seqalign_amd as S
from seqalign_amd import workloads as W
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sizes = [int(x) for x in sys.argv[2:]] or [8, 32, 64, 128, 256, 512]
sc = S.make_scoring({"preset": "default"})
ctx = S.Context(0)
for n in sizes:
    batch = W.dna_nw_150(n, seed=4, length=L)
    res = {0: [], 1: []}
    for r in range(3):
        for v in (0, 1):
            ctx.set_option("nw_dirs", v)
            for _ in range(3): ctx.nw_batch(batch, sc, raw=True)
            ts = []
            for _ in range(11):
                t0 = time.perf_counter(); ctx.nw_batch(batch, sc, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
            res[v].append(float(np.median(ts)))
            if r == 0: print(n, "nw_dirs", v, "launched", ctx.last_call(), flush=True)
    for v in (0, 1):
        print(f"{n} x ({L} x {L}) NW, nw_dirs={v}: " + " ".join("%.3f" % x for x in res[v]) + " ms", flush=True)
