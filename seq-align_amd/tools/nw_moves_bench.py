#!/usr/bin/env python3
"""seqalign_nw_batch end to end on C2 and C5's per-GPU share, one line per variant of the round-4 path:
moves home (nw_moves) x what is read / written in place over PCIe (zero_copy) x the walker (trace_kernel).
    python nw_moves_bench.py [C2|C5|both] [variant ...]      variant = "nw_moves=0,zero_copy=3,trace_kernel=wave"
Prints the median and the minimum of 9 calls after 3 warm-up calls, and checks every variant's strings against the first's."""
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
import numpy as np  # noqa: E402

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
variants = sys.argv[2:] or ["nw_moves=0", "nw_moves=1,zero_copy=0", "nw_moves=1,zero_copy=1", "nw_moves=1,zero_copy=2",
                            "nw_moves=1,zero_copy=3", "nw_moves=1,zero_copy=3,trace_kernel=lane",
                            "nw_moves=1,zero_copy=3,trace_kernel=wave"]
DEFAULTS = {"nw_moves": "1", "zero_copy": "auto", "trace_kernel": "auto", "subbatches": "0", "walk_overlap": "0", "quad": "0"}
batches = []
if which in ("C2", "both"):
    batches.append(("C2", W.dna_nw_150(10000, seed=1)))
if which in ("C5", "both"):
    batches.append(("C5share", W.dna_nw_indexed(875000, 125000, seed=5)))
sc = S.make_scoring({"preset": "default"})
with S.Context(0) as ctx:
    for name, batch in batches:
        want = None
        for v in variants:
            for k, d in DEFAULTS.items():
                ctx.set_option(k, d)
            for kv in v.split(","):
                k, val = kv.split("=")
                ctx.set_option(k, val)
            ts = []
            for it in range(12):
                t0 = time.perf_counter()
                out = ctx.nw_batch(batch, sc, raw=True)
                ts.append((time.perf_counter() - t0) * 1e3)
            got = tuple(np.array(x, copy=True) for x in out[1:])
            if want is None:
                want = got
            same = all(np.array_equal(a, b) for a, b in zip(got, want))
            ts = ts[3:]
            print(f"{name:8s} {v:48s} median {statistics.median(ts):7.3f} ms  min {min(ts):7.3f} ms  "
                  f"{'identical' if same else 'DIFFERENT'}", flush=True)
