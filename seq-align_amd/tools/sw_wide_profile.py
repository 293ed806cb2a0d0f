#!/usr/bin/env python3
"""The kernels of seqalign_sw_batch on reads longer than 512 bp, for rocprofv3 --kernel-trace --stats:
    sw_wide_profile.py [read_len = 700] [max_hits = 4] [calls = 10]     (10 000 reads against 1 000 bp windows)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import seqalign_amd as S
from seqalign_amd import workloads as W
rl = int(sys.argv[1]) if len(sys.argv) > 1 else 700
mh = int(sys.argv[2]) if len(sys.argv) > 2 else 4
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n = 10000
batch = W.dna_sw_read_vs_ref(n, seed=2, read_len=rl, ref_len=1000)
sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
thr = W.default_minscore(sc.match, rl, 1000)
ctx = S.Context(0)
for _ in range(calls): ctx.sw_batch(batch, sc, thr, max_hits=mh, hit_cap=n * mh + 8, raw=True)
print("launched", ctx.last_call(), flush=True)
