#!/usr/bin/env python3
"""The direction byte's LOCAL form (option dirs_local, sa_kernels.h) against the older one on a BASELINE config's host-level call:
    python seq-align_amd/tools/ab_local.py <C2|C3|C4|C5> <dirs_local: 0|1> [calls] [option=value ...]
seqalign_nw_batch (C2, C5's share) / seqalign_sw_batch best hit (C3, C4); prints the call's wall clock and checks the results of the
two forms against each other in the same process; run under rocprofv3 --kernel-trace --stats for the kernels' own durations."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import seqalign_amd as S
from seqalign_amd import workloads as W
from bench import WORKLOADS
name, local = sys.argv[1], sys.argv[2]
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 15
extra = dict(a.split("=", 1) for a in sys.argv[4:])   # further options of the context: key=value
gen, kwargs, n, is_sw, spec, _ = WORKLOADS[name]
n = int(extra.pop("pairs", n))   # (pairs=N: the config's shape, another batch size)
batch = W.dna_nw_indexed(0, n, **kwargs) if name == "C5" else getattr(W, gen)(n, **kwargs)
sc = S.make_scoring(spec)
thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0])) if is_sw else 0
ctx = S.Context(0)
for k_, v_ in extra.items(): ctx.set_option(k_, v_)
call = (lambda: ctx.sw_batch(batch, sc, thr, max_hits=1, hit_cap=n + 8, raw=True)) if is_sw else (lambda: ctx.nw_batch(batch, sc, raw=True))


def digest(r):
    import hashlib
    h = hashlib.sha256()
    for part in (r if isinstance(r, (tuple, list)) else (r,)):
        if isinstance(part, np.ndarray):
            h.update(np.ascontiguousarray(part).tobytes())
        elif isinstance(part, (bytes, bytearray)) or hasattr(part, "_length_"):   # (ctypes arrays: the hits)
            h.update(bytes(part))
        elif isinstance(part, dict):
            for k in sorted(part):
                v = part[k]
                h.update(np.ascontiguousarray(v).tobytes() if isinstance(v, np.ndarray) else repr(v).encode())
        else:
            h.update(repr(part).encode())
    return h.hexdigest()[:16]


ctx.set_option("dirs_local", 1 - int(local)); other = digest(call())
ctx.set_option("dirs_local", local)
for _ in range(5): r = call()
mine = digest(r)
ts = []
for _ in range(calls):
    t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"{name} dirs_local={local} {extra if extra else ''}: median {np.median(ts):.3f} ms min {min(ts):.3f}  same results as dirs_local={1 - int(local)}: {mine == other} ({mine})  launched {sorted(ctx.last_call())}", flush=True)
