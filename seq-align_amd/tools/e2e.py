#!/usr/bin/env python3
"""End-to-end wall time of the host-level batch entry points (host buffers in,
results out): what a caller of seqalign_nw_batch / seqalign_sw_batch sees,
PCIe and host work included.  Not the bench.py metric (that one starts with the
inputs in HBM); reported in DESIGN.md."""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402
from bench import WORKLOADS  # noqa: E402

out = {}
with S.Context(0) as ctx:
    gen, kwargs, n, is_sw, spec, _ = WORKLOADS["C2"]
    batch = getattr(W, gen)(n, **kwargs)
    sc = S.make_scoring(spec)
    for mode in ("device", "host"):
        ctx.set_option("traceback", mode)
        ctx.nw_batch(batch, sc, raw=True)             # warm-up (allocations, pinned staging)
        ts = []
        for _ in range(3):   # raw=True: time the C entry point, not Python tuple building
            t0 = time.perf_counter(); res = ctx.nw_batch(batch, sc, raw=True); ts.append(time.perf_counter() - t0)
        out[f"nw_batch_C2_traceback_{mode}"] = dict(seconds=min(ts), gcups=batch.cells() / min(ts) / 1e9,
                                                    pairs_per_s=n / min(ts))
    M, A, B, _, _ = ctx.fill_batch(batch, sc, 0)      # first call also faults the pages in
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ctx.fill_batch(batch, sc, 0, out=(M, A, B)); ts.append(time.perf_counter() - t0)
    out["fill_batch_C2_matrices_to_host"] = dict(seconds=min(ts), gcups=batch.cells() / min(ts) / 1e9,
                                                 GBps=12 * float(batch.matrix_cells().sum()) / min(ts) / 1e9)
    del M, A, B
    ctx.set_option("traceback", "device")
    for name in ("C3", "C4"):
        gen, kwargs, n, is_sw, spec, _ = WORKLOADS[name]
        sc = S.make_scoring(spec)
        # best hit only, full config: fill + reduction + traceback all on the device
        batch = getattr(W, gen)(n, **kwargs)
        thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
        ctx.sw_batch(batch, sc, thr, max_hits=1, hit_cap=n + 8, raw=True)
        ts = []
        for _ in range(3):   # raw=True: time the C entry point, not Python dict building
            t0 = time.perf_counter(); nh = ctx.sw_batch(batch, sc, thr, max_hits=1, hit_cap=n + 8, raw=True)[0]
            ts.append(time.perf_counter() - t0)
        out[f"sw_batch_{name}_best_hit_device"] = dict(seconds=min(ts), gcups=batch.cells() / min(ts) / 1e9, hits=nh)
        # up to 4 hits per pair, enumerated on the device (full config)
        ctx.sw_batch(batch, sc, thr, max_hits=4, hit_cap=4 * n + 8, raw=True)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); nh = ctx.sw_batch(batch, sc, thr, max_hits=4, hit_cap=4 * n + 8, raw=True)[0]
            ts.append(time.perf_counter() - t0)
        out[f"sw_batch_{name}_4hits_device"] = dict(seconds=min(ts), gcups=batch.cells() / min(ts) / 1e9, hits=nh)
        # the same through the host path (candidates + matrices over PCIe), a tenth of the config
        ctx.set_option("traceback", "host")
        batch = getattr(W, gen)(n // 10, **kwargs)
        ctx.sw_batch(batch, sc, thr, max_hits=4, raw=True)
        t0 = time.perf_counter(); nh = ctx.sw_batch(batch, sc, thr, max_hits=4, raw=True)[0]; t1 = time.perf_counter()
        out[f"sw_batch_{name}_tenth_4hits_host"] = dict(seconds=t1 - t0, gcups=batch.cells() / (t1 - t0) / 1e9, hits=nh)
        ctx.set_option("traceback", "device")
print(json.dumps(out, indent=1))
