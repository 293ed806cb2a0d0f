#!/usr/bin/env python3
"""What the ORIENTATION of a pair costs the best-hit fill: BASELINE configs[2]'s pairs as they are (rows of 151 columns x 1 001 rows: seq_a = the
read) and with the two sequences swapped (rows of 1 001 columns x 151 rows) -- same cells, same scoring; the swapped call's ALIGNMENTS are
not the reference's (gap_a / gap_b and the tie order swap roles), this only asks what a transposed kernel would be worth.  Kernel means
under rocprofv3 --kernel-trace --stats; prints the calls' wall clock."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np
import torch  # noqa: F401
import seqalign_amd as S
from seqalign_amd import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
b0 = W.dna_sw_read_vs_ref(n, seed=2)
pairs = [(b0.seq_a(p), b0.seq_b(p)) for p in range(n)]
swapped = W.from_pairs([(b, a) for a, b in pairs])
sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
thr = W.default_minscore(sc.match, 150, 1000)
ctx = S.Context(0)
for name, batch in (("as they are (151-column rows)", b0), ("swapped (1 001-column rows)", swapped)):
    call = lambda: ctx.sw_batch(batch, sc, thr, max_hits=1, hit_cap=n + 8, raw=True)
    for _ in range(4): call()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: median {np.median(ts):.3f} ms  launched {sorted(ctx.last_call())}", flush=True)
