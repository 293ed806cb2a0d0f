#!/usr/bin/env python3
"""seqalign_nw_batch on C2 (or seqalign_sw_batch best hit on C3 / C4) with the walker and the way its moves travel home forced:
    python seq-align_amd/tools/walk_zc_ab.py <workload> <trace_kernel: auto|lane|wave> <zero_copy: auto|0|1|2|3> [walk_group]
prints the call's wall clock; run under rocprofv3 --kernel-trace --stats for the kernels' own durations."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import seqalign_amd as S
from seqalign_amd import workloads as W
from bench import WORKLOADS
name, walker, zc = sys.argv[1], sys.argv[2], sys.argv[3]
grp = int(sys.argv[4]) if len(sys.argv) > 4 else 0
gen, kwargs, n, is_sw, spec, _ = WORKLOADS[name]
batch = getattr(W, gen)(n, **kwargs)
sc = S.make_scoring(spec)
thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0])) if is_sw else 0
ctx = S.Context(0)
ctx.set_option("trace_kernel", walker); ctx.set_option("zero_copy", zc); ctx.set_option("walk_group", grp)
call = (lambda: ctx.sw_batch(batch, sc, thr, max_hits=1, hit_cap=n + 8, raw=True)) if is_sw else (lambda: ctx.nw_batch(batch, sc, raw=True))
for _ in range(5): call()
ts = []
for _ in range(15):
    t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"{name} trace_kernel={walker} zero_copy={zc} walk_group={grp}: median {np.median(ts):.3f} ms min {min(ts):.3f}  launched {sorted(ctx.last_call())}", flush=True)
