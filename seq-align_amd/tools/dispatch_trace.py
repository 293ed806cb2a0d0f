#!/usr/bin/env python3
"""Where and when does each wave of the stream kernel run?  Needs the experiment
build  make -C seq-align_amd exp EXPFLAGS=-DSA_EXP_TRACE  (the kernel then writes
xcc/se/cu/simd and start/end timestamps into status[] instead of the error index).

    SEQALIGN_LIB=seq-align_amd/lib/libseqalign_hip_exp.so python seq-align_amd/tools/dispatch_trace.py 2048 10000"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402
from bench import WORKLOADS  # noqa: E402

gen, kwargs, per_gpu, is_sw, spec, _ = WORKLOADS["C2"]
ctx = S.Context(0)
h = ctx.upload_scoring(S.make_scoring(spec), is_sw)
out = {}
for n in [int(x) for x in sys.argv[1:]] or [10000]:
    batch = getattr(W, gen)(n, **kwargs)
    db = S.DeviceBatch(batch, 0)
    for _ in range(30):
        db.fill(ctx, h, S.KERNEL_STREAM)
    torch.cuda.synchronize()
    st = db.status.cpu().numpy().view(np.uint64)
    xcc = (st >> np.uint64(60)).astype(np.int64)
    hw = ((st >> np.uint64(44)) & np.uint64(0xffff)).astype(np.int64)
    t0 = ((st >> np.uint64(22)) & np.uint64(0x3fffff)).astype(np.int64)
    t1 = (st & np.uint64(0x3fffff)).astype(np.int64)
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    base = t0.min()
    t0 = (t0 - base) & 0x3fffff
    t1 = (t1 - base) & 0x3fffff
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = cu_key * 4 + simd
    ncu = len(np.unique(cu_key))
    per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1])
    per_simd = np.bincount(np.unique(simd_key, return_inverse=True)[1])
    dur = (t1 - t0) * 0.01   # us
    # concurrency profile: waves alive at each 5 us
    T = t1.max() * 0.01
    grid = np.arange(0, T, 5.0)
    alive = [(int(((t0 * 0.01 <= g) & (t1 * 0.01 > g)).sum())) for g in grid]
    # max waves alive at once on any CU
    ev = sorted([(a, 1, k) for a, k in zip(t0, cu_key)] + [(b, -1, k) for b, k in zip(t1, cu_key)])
    cur, peak = {}, 0
    for _, d, k in ev:
        cur[k] = cur.get(k, 0) + d
        peak = max(peak, cur[k])
    rec = dict(pairs=n, kernel_us=float(T), cus_used=int(ncu), waves_per_cu_min=int(per_cu.min()),
               waves_per_cu_max=int(per_cu.max()), waves_per_simd_min=int(per_simd.min()),
               waves_per_simd_max=int(per_simd.max()), peak_waves_alive_on_a_cu=int(peak),
               wave_us_min=float(dur.min()), wave_us_median=float(np.median(dur)), wave_us_max=float(dur.max()),
               start_us_pcts=[float(x) for x in np.percentile(t0 * 0.01, [0, 25, 50, 75, 90, 100])],
               alive_every_5us=alive)
    out[n] = rec
    print(json.dumps(rec), flush=True)
