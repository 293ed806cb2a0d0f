#!/usr/bin/env python3
"""BLOSUM62 300 x 300 (BASELINE configs[3]'s shape) by batch size: seqalign_sw_batch(max_hits = 1 and 4) on 4 000 .. 32 000 pairs, for
rocprofv3 --kernel-trace (profiles/scripts/c4_by_batch_size.sh turns the trace into per-size kernel times and issue fractions).
The question: is the table fills' 0.55 of VALU issue at C4's 4 000 pairs a property of the kernel or of the launch's size?"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import seqalign_amd as S
from seqalign_amd import workloads as W
sc = S.make_scoring({"preset": "BLOSUM62"})
ctx = S.Context(0)
for n in (4000, 8000, 16000, 32000):
    batch = W.protein_sw_300(n, seed=3)
    thr = W.default_minscore(sc.match, 300, 300)
    for hits in (1, 4):
        ts = []
        for it in range(4):
            t0 = time.perf_counter()
            ctx.sw_batch(batch, sc, thr, max_hits=hits, hit_cap=hits * n + 8, raw=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"pairs {n} max_hits {hits}: call {min(ts[1:]):.3f} ms", flush=True)
