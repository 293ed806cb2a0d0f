#!/usr/bin/env python3
"""Stage breakdown of seqalign_nw_batch (option timing=1: laps on stderr) on C2 and on C5's per-GPU share."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402

with S.Context(0) as ctx:
    ctx.set_option("timing", 1)
    sc = S.make_scoring({"preset": "default"})
    for name, batch in (("C2", W.dna_nw_150(10000, seed=1)), ("C5share", W.dna_nw_indexed(875000, 125000, seed=5))):
        ctx.nw_batch(batch, sc, raw=True)
        for rep in range(3):
            sys.stderr.write(f"---- {name} call {rep}\n")
            t0 = time.perf_counter()
            ctx.nw_batch(batch, sc, raw=True)
            sys.stderr.write(f"---- {name} wall {(time.perf_counter() - t0) * 1e3:.3f} ms\n")
