#!/usr/bin/env python3
"""seqalign_sw_batch end to end on C3 / C4, best hit and up to 4 hits, one line per option variant (same box, same process):
    python sw_moves_bench.py [C3|C4|both] [variant ...]      variant = "nw_moves=0" ...
Median / minimum of 9 calls after 3 warm-up calls; every variant's hits are compared with the first's."""
import ctypes as C
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402
from bench import WORKLOADS  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
variants = sys.argv[2:] or ["nw_moves=0", "nw_moves=1"]
DEFAULTS = {"nw_moves": "1", "trace_kernel": "auto", "pack16": "1", "sweep_dirs": "1", "timing": "0", "sweep_ev": "1", "subbatches": "0", "quad": "0"}
with S.Context(0) as ctx:
    for name in (["C3", "C4"] if which == "both" else [which]):
        gen, kwargs, n, is_sw, spec, _ = WORKLOADS[name]
        batch = getattr(W, gen)(n, **kwargs)
        sc = S.make_scoring(spec)
        thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
        for max_hits in (1, 4):
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
                    out = ctx.sw_batch(batch, sc, thr, max_hits=max_hits, hit_cap=max_hits * n + 8, raw=True)
                    ts.append((time.perf_counter() - t0) * 1e3)
                nh = int(out[0])
                raw = C.string_at(out[1], nh * C.sizeof(S.SwHit))
                used = (out[1][nh - 1].str_off + out[1][nh - 1].length + 1) if nh else 0
                got = (nh, raw, bytes(out[2][:used]), bytes(out[3][:used]))
                if want is None:
                    want = got
                ts = ts[3:]
                print(f"{name} max_hits {max_hits}  {v:40s} median {statistics.median(ts):7.3f} ms  min {min(ts):7.3f} ms  hits {nh}  "
                      f"{'identical' if got == want else 'DIFFERENT'}  {ctx.last_call()}", flush=True)
