#!/usr/bin/env python3
"""Drive seqalign_nw_batch for rocprofv3 / host-side timing of its stages
(SEQALIGN_TIMING=1 prints the library's own stage timers)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402
from bench import WORKLOADS  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
gen, kwargs, _, is_sw, spec, _ = WORKLOADS["C2"]
batch = getattr(W, gen)(n, **kwargs)
sc = S.make_scoring(spec)
ctx = S.Context(0)
for it in range(6):
    t0 = time.perf_counter()
    res = ctx.nw_batch(batch, sc, raw=True)
    print("nw_batch", n, "pairs %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
# (for profiles/e2e_roofline_summarise.py: the walkers' latency bound needs the walks' lengths)
print("walks %d mean_steps %.1f" % (n, float(res[3].mean())), flush=True)
