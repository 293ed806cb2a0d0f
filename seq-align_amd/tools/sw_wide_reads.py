#!/usr/bin/env python3
"""seqalign_sw_batch on reads longer than 512 bp (10 000 reads of 700 bp against 1 000 bp windows), the direction-byte path
against the three-matrix path these rows took before round 5, alternating in one process:
  sw_wide_reads.py [read_len] 1     best hit: the packed fill (pack16 = 1) against pack16 = 0
  sw_wide_reads.py [read_len] 4     up to 4 hits: match_scores + directions + the one-word sweep (sweep_ev = 1) against sweep_ev = 0"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np
import seqalign_amd as S
from seqalign_amd import workloads as W
n, rl = 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 700
mh = int(sys.argv[2]) if len(sys.argv) > 2 else 1
opt = "pack16" if mh == 1 else "sweep_ev"
cap = n * mh + 8
batch = W.dna_sw_read_vs_ref(n, seed=2, read_len=rl, ref_len=1000)
sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
thr = W.default_minscore(sc.match, rl, 1000)
ctx = S.Context(0)
res = {0: [], 1: []}
for r in range(4):
    for pk in (0, 1):
        ctx.set_option(opt, pk)
        for _ in range(3): ctx.sw_batch(batch, sc, thr, max_hits=mh, hit_cap=cap, raw=True)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); ctx.sw_batch(batch, sc, thr, max_hits=mh, hit_cap=cap, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
        res[pk].append(float(np.median(ts)))
        if r == 0: print(opt, pk, "launched", ctx.last_call(), flush=True)
for pk in (0, 1):
    print(f"10 000 x ({rl} x 1000) up to {mh} hit(s), {opt}={pk}: " + " ".join("%.3f" % x for x in res[pk]) + " ms", flush=True)
