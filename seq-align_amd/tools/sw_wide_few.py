#!/usr/bin/env python3
"""FEW wide pairs, up to 4 hits: the direction-byte path with one wave per pair (sweep_mode = pair) against what the batch takes by
default below 128 pairs (three matrices, one wave per 256-column strip; below 1 024 pairs until this record), alternating in one process.
    sw_wide_few.py [read_len = 700] [pairs ...]"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np
import seqalign_amd as S
from seqalign_amd import workloads as W
rl = int(sys.argv[1]) if len(sys.argv) > 1 else 700
sizes = [int(x) for x in sys.argv[2:]] or [64, 128, 256, 512, 1000]
sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
thr = W.default_minscore(sc.match, rl, 1000)
ctx = S.Context(0)
for n in sizes:
    batch = W.dna_sw_read_vs_ref(n, seed=2, read_len=rl, ref_len=1000)
    modes = os.environ.get("SW_MODES", "auto,pair").split(",")   # (strips: three matrices + one wave per strip whatever the size)
    res = {m: [] for m in modes}
    for r in range(3):
        for mode in modes:
            ctx.set_option("sweep_mode", mode)
            for _ in range(3): ctx.sw_batch(batch, sc, thr, max_hits=4, hit_cap=4 * n + 8, raw=True)
            ts = []
            for _ in range(9):
                t0 = time.perf_counter(); ctx.sw_batch(batch, sc, thr, max_hits=4, hit_cap=4 * n + 8, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
            res[mode].append(float(np.median(ts)))
            if r == 0: print(n, mode, "launched", ctx.last_call(), flush=True)
    for mode in modes:
        print(f"{n} x ({rl} x 1000) up to 4 hits, sweep_mode={mode}: " + " ".join("%.3f" % x for x in res[mode]) + " ms", flush=True)
