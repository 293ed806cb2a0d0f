#!/usr/bin/env python3
"""The packed fills (two / four pairs per wave) against the one-pair direction fills BELOW kPackedFillMinPairs = 2 048 pairs:
seqalign_nw_batch on C2's shape and seqalign_sw_batch(max_hits = 4) on C3's / C4's, default (pack16 = 1) against forced (pack16 = 2),
alternating in one process.    pack_by_batch_size.py [nw | C3 | C4] [pairs ...]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np
import seqalign_amd as S
from seqalign_amd import workloads as W
kind = sys.argv[1] if len(sys.argv) > 1 else "nw"
sizes = [int(x) for x in sys.argv[2:]] or [256, 512, 1024, 1536, 2047]
ctx = S.Context(0)
for n in sizes:
    if kind == "nw":
        batch = W.dna_nw_150(n, seed=1); sc = S.make_scoring({"preset": "default"})
        call = lambda: ctx.nw_batch(batch, sc, raw=True)
    else:
        if kind == "C4":
            batch = W.protein_sw_300(n, seed=3); sc = S.make_scoring({"preset": "BLOSUM62"})
        else:
            batch = W.dna_sw_read_vs_ref(n, seed=2); sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
        thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
        call = lambda: ctx.sw_batch(batch, sc, thr, max_hits=4, hit_cap=4 * n + 8, raw=True)
    res = {1: [], 2: []}
    for r in range(3):
        for pk in (1, 2):
            ctx.set_option("pack16", pk)
            for _ in range(3): call()
            ts = []
            for _ in range(15):
                t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
            res[pk].append(float(np.median(ts)))
            if r == 0: print(kind, n, "pack16", pk, "launched", ctx.last_call(), flush=True)
    for pk in (1, 2):
        print(f"{kind} {n} pairs, pack16={pk}: " + " ".join("%.3f" % x for x in res[pk]) + " ms", flush=True)
