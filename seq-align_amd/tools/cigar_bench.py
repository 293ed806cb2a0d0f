#!/usr/bin/env python3
"""seqalign_nw_batch against seqalign_nw_batch_cigar on BASELINE configs[1] (10 000 pairs, 150 x 150): the call's wall clock with the
alignments coming home as two gapped strings per pair, as CIGAR in worst-case slots and in slots of 256 bytes; then the CIGAR call's stage laps."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np
import torch  # noqa: F401
import seqalign_amd as S
from seqalign_amd import workloads as W
batch = W.dna_nw_150(10000, seed=1); sc = S.make_scoring({"preset": "default"})
ctx = S.Context(0)


def t(f, n=25):
    for _ in range(5): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return "median %.3f ms min %.3f" % (np.median(ts), min(ts))


print("strings          ", t(lambda: ctx.nw_batch(batch, sc, raw=True)), flush=True)
print("cigar, worst case", t(lambda: ctx.nw_batch_cigar(batch, sc, 1, raw=True)), flush=True)
print("cigar, slot 256  ", t(lambda: ctx.nw_batch_cigar(batch, sc, 1, slot=256, raw=True)), flush=True)
ctx.set_option("timing", 1)
ctx.nw_batch_cigar(batch, sc, 1, raw=True)
