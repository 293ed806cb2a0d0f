#!/usr/bin/env python3
"""seqalign_arenas_alloc when the device is not empty, and what a walk costs afterwards:
  (1) with 150 GB held by somebody else (a torch tensor): C2-sized arenas, default options and arena_scan_gib = 24 -- the walk's
      record (quality, candidates, memory held at the end);
  (2) the transient after a walk of 24 / 64 / 160 GiB: seqalign_nw_batch on C2 timed for 3 s after the arenas came back
      (the driver wipes released VRAM in the background; profiles/r03/r03_after_placement_transient.txt)."""
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402

NB = 4 * 10000 * 22801 // 4096 * 4096 + 4096       # one C2 arena


def alloc(ctx, tag):
    ptrs, q = (C.c_void_p * 3)(), C.c_float(-1)
    t0 = time.perf_counter()
    S._check(S.lib().seqalign_arenas_alloc(ctx._h, C.c_uint64(NB), ptrs, C.byref(q)), "alloc")
    dt = time.perf_counter() - t0
    info = S.ArenaInfo()
    S.lib().seqalign_arenas_info(ctx._h, ptrs, C.byref(info))
    d = info.as_dict()
    print(f"{tag}: quality {d['quality']} scanned {d['scanned_gib']} GiB tries {d['tries']} depth B {d['depth_gib']} A {d['depth_a_gib']}  {dt * 1e3:.0f} ms", flush=True)
    return ptrs


torch.cuda.set_device(0)
batch = W.dna_nw_150(10000, seed=1)
sc = S.make_scoring({"preset": "default"})
with S.Context(0) as ctx:
    for _ in range(5):
        ctx.nw_batch(batch, sc, raw=True)
    free0 = torch.cuda.mem_get_info(0)[0]
    print(f"free {free0 / 2**30:.1f} GiB", flush=True)
    hog = torch.empty(150 * 10**9, dtype=torch.uint8, device="cuda")
    print(f"holding 150 GB, free {torch.cuda.mem_get_info(0)[0] / 2**30:.1f} GiB", flush=True)
    for scan in (160, 24):
        ctx.set_option("arena_scan_gib", scan)
        p = alloc(ctx, f"under pressure, arena_scan_gib={scan}")
        S.lib().seqalign_arenas_free(ctx._h, p)
        time.sleep(3)
    del hog
    torch.cuda.empty_cache()
    time.sleep(4)
    for scan in (24, 64, 160):
        ctx.set_option("arena_scan_gib", scan)
        ctx.set_option("arena_quality", 1.4)          # never satisfied: the walk uses its whole budget
        p = alloc(ctx, f"empty device, whole budget of {scan} GiB")
        t_rel = time.perf_counter()
        ts = []
        while time.perf_counter() - t_rel < 3.0:
            t0 = time.perf_counter(); ctx.nw_batch(batch, sc, raw=True); ts.append(((time.perf_counter() - t_rel), (time.perf_counter() - t0) * 1e3))
        import statistics
        for lo, hi in ((0, 0.25), (0.25, 0.5), (0.5, 1), (1, 2), (2, 3)):
            sel = [ms for t, ms in ts if lo <= t < hi]
            if sel:
                print(f"   {lo:4.2f}-{hi:4.2f} s after the walk: nw_batch median {statistics.median(sel):.3f} ms ({len(sel)} calls)", flush=True)
        S.lib().seqalign_arenas_free(ctx._h, p)
        time.sleep(4)
