#!/usr/bin/env python3
"""Drive seqalign_sw_batch (device multi-hit enumeration) for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python seq-align_amd/tools/sw_enum_profile.py C3 4"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
max_hits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
gen, kwargs, n, is_sw, spec, _ = WORKLOADS[name]
batch = getattr(W, gen)(n, **kwargs)
sc = S.make_scoring(spec)
thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
ctx = S.Context(0)
for it in range(3):
    t0 = time.perf_counter()
    res = ctx.sw_batch(batch, sc, thr, max_hits=max_hits, hit_cap=min(max_hits, 20) * n + 8, raw=True)
    nh = res[0]
    print(name, "max_hits", max_hits, "hits", nh, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
# (for profiles/e2e_roofline_summarise.py: the walkers' latency bound needs the walks' lengths)
try:
    hits = res[1]
    print("walks %d mean_steps %.1f" % (nh, float(np.mean([h.length for h in hits[:nh]])) if nh else 0.0), flush=True)
except Exception as e:      # the raw result's layout is the driver's business: no bound then
    print("walks: no lengths (%s)" % e, flush=True)
