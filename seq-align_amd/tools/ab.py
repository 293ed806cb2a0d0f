#!/usr/bin/env python3
"""In-process interleaved A/B timing of fill-kernel variants (guide rule 24:
cross-process deltas are noise; interleave variants in ONE process and report
the distribution).

    python seq-align_amd/tools/ab.py --workload C2 --rounds 7 --launches 10 \
        "stream" "stream:SEQALIGN_LDS_PAD=16384" "rowscan"

A variant is  kernel[:ENV=VALUE[,ENV=VALUE...]]; the env vars are set around the
launches (the launchers read them at launch time).  To compare two BUILDS run
the tool once per build with SEQALIGN_LIB set and alternate the invocations."""
import argparse
import json
import os
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

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--workload", default="C2")
ap.add_argument("--pairs", type=int, default=0)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--launches", type=int, default=10)
ap.add_argument("--skew", type=int, default=0, help="extra ints between the three arenas (layout experiments; packed only)")
ap.add_argument("--placement", default="spread", choices=["spread", "packed"])
args = ap.parse_args()

gen, kwargs, per_gpu, is_sw, spec, _ = WORKLOADS[args.workload]
batch = getattr(W, gen)(args.pairs or per_gpu, **kwargs)
os.environ["SEQALIGN_ARENA_SKEW"] = str(args.skew)
ctx = S.Context(0)
h = ctx.upload_scoring(S.make_scoring(spec), is_sw)
db = S.DeviceBatch(batch, 0, placement="packed" if args.skew else args.placement, ctx=ctx)
KID = {"wavefront": S.KERNEL_WAVEFRONT, "rowscan": S.KERNEL_ROWSCAN, "stream": S.KERNEL_STREAM,
       "memset": -1}   # memset = torch fill_ of the same three arenas: this box's write ceiling


def time_memset(repeats):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(repeats + 1)]
    with torch.cuda.stream(db.stream):
        for i in range(repeats):
            ev[i].record(db.stream)
            db.M.fill_(7); db.A.fill_(7); db.B.fill_(7)
        ev[repeats].record(db.stream)
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(repeats)]


parsed = []
for v in args.variants:
    name, _, envs = v.partition(":")
    env = dict(e.split("=", 1) for e in envs.split(",") if e)
    parsed.append((v, KID[name], env))
res = {v: [] for v, _, _ in parsed}
for r in range(args.rounds + 1):
    for v, kid, env in parsed:
        # variant options: SEQALIGN_<KEY>=value or key=value, set on the context for this variant's launches
        keys = {k: (k[len("SEQALIGN_"):].lower() if k.startswith("SEQALIGN_") else k) for k in env}
        for k, v_ in env.items():
            ctx.set_option(keys[k], v_)
        ms = time_memset(args.launches) if kid < 0 else db.time_fill_ms(ctx, h, kid, args.launches)
        for k in env:
            ctx.set_option(keys[k], S.OPTION_DEFAULTS[keys[k]])
        if r:   # round 0 = warm-up
            res[v].append(float(np.median(ms)))
alg = db.algorithmic_bytes()
for v, xs in res.items():
    med, lo = float(np.median(xs)), float(min(xs))
    print(f"{v:48s} median {med:.4f} ms  min {lo:.4f} ms  {batch.cells() / med / 1e6:7.1f} GCUPS  "
          f"{alg / med / 1e6:7.0f} GB/s  frac {alg / med / 1e6 / 8000:.3f}")
