#!/usr/bin/env python3
"""The tile walker by walks per wave (option walk_group: 1 = one wave per walk, round 5; 4 / 8 = in lockstep, round 6): results equal,
and the wall clock of seqalign_nw_batch (C2) / seqalign_sw_batch best hit (C3, C4), alternating in one process.  Kernel times:
run under rocprofv3 --kernel-trace --stats."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import seqalign_amd as S
from seqalign_amd import workloads as W
from bench import WORKLOADS
ctx = S.Context(0)
for name in (sys.argv[1:] or ["C2", "C3", "C4"]):
    gen, kwargs, n, is_sw, spec, _ = WORKLOADS[name]
    batch = getattr(W, gen)(n, **kwargs)
    sc = S.make_scoring(spec)
    thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0])) if is_sw else 0
    call = (lambda: ctx.sw_batch(batch, sc, thr, max_hits=1, hit_cap=n + 8, raw=True)) if is_sw else (lambda: ctx.nw_batch(batch, sc, raw=True))
    ref = None
    for rep in range(3):
        for grp in (1, 4, 8):
            ctx.set_option("walk_group", grp)
            for _ in range(3):
                call()
            ts = []
            for _ in range(9):
                t0 = time.perf_counter(); r = call(); ts.append((time.perf_counter() - t0) * 1e3)
            if is_sw:
                nh, hits, oa, ob = r
                digest = (nh, bytes(memoryview(hits)[: nh * 40]) if False else tuple((hits[k].score, hits[k].pos_a, hits[k].pos_b, hits[k].length) for k in range(0, nh, max(1, nh // 200))), int(oa[: 1 << 20].sum()))
            else:
                digest = tuple(int(x.astype(np.uint64).sum()) for x in r[1:])
            if ref is None: ref = digest
            print(f"{name} walk_group {grp}: median {np.median(ts):.3f} ms  min {min(ts):.3f}  same_as_first {digest == ref}  launched {sorted(ctx.last_call())}", flush=True)
