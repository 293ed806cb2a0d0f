#!/usr/bin/env python3
"""seqalign_sw_batch(max_hits = 1) by batch size: what runs below the packed fills' 2 048 pairs (pack16 = 1, the default) against the
packed best-hit fill forced on (pack16 = 2), alternating in one process.   sw_best_few.py [C3 | C4 | wide | wide<read length>] [pairs ...]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np
import seqalign_amd as S
from seqalign_amd import workloads as W
kind = sys.argv[1] if len(sys.argv) > 1 else "C3"
sizes = [int(x) for x in sys.argv[2:]] or [128, 256, 512, 1024, 2047]
ctx = S.Context(0)
for n in sizes:
    if kind == "C4":
        batch = W.protein_sw_300(n, seed=3); sc = S.make_scoring({"preset": "BLOSUM62"})
    else:
        batch = W.dna_sw_read_vs_ref(n, seed=2, read_len=int(kind[4:] or 700) if kind.startswith("wide") else 150, ref_len=1000)
        sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
    res = {1: [], 2: []}
    for r in range(3):
        for pk in (1, 2):
            ctx.set_option("pack16", pk)
            for _ in range(3): ctx.sw_batch(batch, sc, thr, max_hits=1, hit_cap=n + 8, raw=True)
            ts = []
            for _ in range(15):
                t0 = time.perf_counter(); ctx.sw_batch(batch, sc, thr, max_hits=1, hit_cap=n + 8, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
            res[pk].append(float(np.median(ts)))
            if r == 0: print(kind, n, "pack16", pk, "launched", ctx.last_call(), flush=True)
    for pk in (1, 2):
        print(f"{kind} {n} pairs, best hit, pack16={pk}: " + " ".join("%.3f" % x for x in res[pk]) + " ms", flush=True)
