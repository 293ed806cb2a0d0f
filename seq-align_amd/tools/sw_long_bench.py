#!/usr/bin/env python3
"""seqalign_sw_batch (up to 4 hits) on few LONG pairs (both sequences thousands of characters)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np, torch
import seqalign_amd as S
from seqalign_amd import workloads as W
rng = W.Rng(9)
def rand(n): return bytes(b"ACGT"[i] for i in rng.below(4, n))
def related(n):
    a = bytearray(rand(n)); b = bytearray(a)
    for i in range(0, n, 23): b[i] = b"ACGT"[(b[i] + 1) % 4]
    return bytes(a), bytes(b[: n - n // 10])
sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
ctx = S.Context(0)
for n_len, n in ((2000, 64), (5000, 8), (10000, 2)):
    batch = W.from_pairs([related(n_len) for _ in range(n)])
    for max_hits in (1, 4):
        for it in range(2):
            t0 = time.perf_counter()
            nh = ctx.sw_batch(batch, sc, n_len // 2, max_hits=max_hits, hit_cap=8 * n + 8, raw=True)[0]
            dt = (time.perf_counter() - t0) * 1e3
        print("%d x %d^2 max_hits %d hits %d  %.2f ms" % (n, n_len, max_hits, nh, dt), flush=True)
