#!/usr/bin/env python3
"""A/B of one context option on seqalign_nw_batch, alternating in one process:  ab_option.py <option> <v1,v2,..> [C2|C5share] [rounds]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np
import seqalign_amd as S
from seqalign_amd import workloads as W
opt, vals = sys.argv[1], sys.argv[2].split(",")
wl = sys.argv[3] if len(sys.argv) > 3 else "C2"
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
batch = (W.dna_nw_150(10000, seed=1) if wl == "C2" else W.dna_nw_150(1000, seed=4, length=1000) if wl == "L1000" else
         W.dna_nw_150(4000, seed=4, length=700) if wl == "L700" else W.dna_nw_indexed(0, 125000, seed=5))
sc = S.make_scoring({"preset": "default"})
ctx = S.Context(0)
for _ in range(5): ctx.nw_batch(batch, sc, raw=True)
res = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        ctx.set_option(opt, v)
        for _ in range(3): ctx.nw_batch(batch, sc, raw=True)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); ctx.nw_batch(batch, sc, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
        res[v].append(float(np.median(ts)))
for v in vals:
    print(f"{wl} {opt}={v}: median of medians {np.median(res[v]):.4f} ms   rounds: " + " ".join("%.3f" % x for x in res[v]), flush=True)
