#!/usr/bin/env python3
"""seqalign_sw_batch (up to 4 hits per pair) on WIDE pairs -- a long seq_a against short reads: the sweep works in
segments that follow the walks (sa_sw_sweep.hip), with the winners of two rows in LDS / in HBM.  DESIGN.md 3.6."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np, torch
import seqalign_amd as S
from seqalign_amd import workloads as W
rng = W.Rng(5)
def rand(n): return bytes(b"ACGT"[i] for i in rng.below(4, n))
def planted(la, lb):
    ref = rand(la); cut = int(rng.below(max(1, la - lb + 1), 1)[0])
    read = bytearray(ref[cut:cut + lb])
    for i in range(0, len(read), 17): read[i] = b"ACGT"[(read[i] + 1) % 4]
    return ref, bytes(read)
sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
ctx = S.Context(0)
for la, lb, n in ((1000, 150, 2000), (1500, 300, 500), (5000, 150, 300), (20000, 200, 40)):
    batch = W.from_pairs([planted(la, lb) for _ in range(n)])
    thr = lb  # half of a perfect read's score
    for it in range(3):
        t0 = time.perf_counter()
        nh = ctx.sw_batch(batch, sc, thr, max_hits=4, hit_cap=8 * n + 8, raw=True)[0]
        dt = (time.perf_counter() - t0) * 1e3
    print("a=%d b=%d n=%d hits %d  %.2f ms" % (la, lb, n, nh, dt), flush=True)
