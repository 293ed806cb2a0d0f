#!/usr/bin/env python3
"""seqalign_nw_batch on a batch whose pairs are mostly 150 x 150 with every tenth pair trimmed: option pack16 = 0 / 1 (wall clock, ms)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np, torch
import seqalign_amd as S
from seqalign_amd import workloads as W
rng = W.Rng(99)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dna = np.frombuffer(b"ACGT", np.uint8)
a = dna[rng.below(4, n * 150).astype(np.int64)].reshape(n, 150)
b = dna[rng.below(4, n * 150).astype(np.int64)].reshape(n, 150)
pairs = []
for k in range(n):
    la = lb = 150
    if k % 10 == 7:
        la, lb = int(60 + rng.below(90, 1)[0]), int(60 + rng.below(90, 1)[0])
    pairs.append((a[k, :la].tobytes(), b[k, :lb].tobytes()))
batch = W.from_pairs(pairs)
sc = S.make_scoring({"preset": "default"})
ctx = S.Context(0)
ref = None
for pk in (0, 1, 0, 1):
    ctx.set_option("pack16", pk)
    ts = []
    for it in range(8):
        t0 = time.perf_counter(); r = ctx.nw_batch(batch, sc, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
    out = [x.copy() for x in r]
    if ref is None: ref = out
    same = all(np.array_equal(x, y) for x, y in zip(ref, out))
    print("mostly 150 x 150, %d pairs, pack16=%d: " % (n, pk) + " ".join("%.3f" % t for t in ts[3:]) + " ms", "identical" if same else "DIFFERENT", flush=True)
