#!/usr/bin/env python3
"""Long-running differential fuzz: random scorings (penalties, the five flags, case
sensitivity, wildcards, asymmetric mutations), random ragged batches up to `--max-len`
columns, EVERY fill kernel, NW and SW, against the oracle (bit-exact).  Not part of the
test suite (minutes); run it after touching a kernel:

    python seq-align_amd/tools/fuzz.py --seconds 300 --max-len 1400
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import orclib as O  # noqa: E402
import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--max-len", type=int, default=1400)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = W.Rng(args.seed)
ctx = S.Context(0)
KERNELS = [S.KERNEL_WAVEFRONT, S.KERNEL_ROWSCAN, S.KERNEL_STREAM, S.KERNEL_STRIPS, S.KERNEL_WGSTREAM, S.KERNEL_AUTO]
t_end = time.time() + args.seconds
trials = pairs_checked = 0
while time.time() < t_end:
    v = rng.below(1 << 20, 14).astype(int)
    flags = [int(v[0] >> k) & 1 for k in range(5)]
    if v[13] % 3 == 0:
        flags = [0, 0, 0, 0, 0]                   # plain scorings: a third of the draws, not 1 in 32
    match, mismatch = int(v[1] % 6), -int(v[2] % 7)
    go, ge = -int(v[3] % 12), -int(v[4] % 4)
    if v[13] % 7 == 0:
        ge = int(1 + v[13] % 2)                   # gap_extend > 0 (legal upstream): the row sweeps' trend from the right end
    if v[11] % 9 == 0:
        go = int(v[11] % 4)                       # gap_open >= 0: the GENERAL path of the row-sweep kernels
    if flags[2] and flags[3]:
        mismatch = min(mismatch, go + ge)
    spec = {"init": [match, mismatch, go, ge, *flags, int(v[5] & 1)], "wildcards": [], "mutations": []}
    if v[6] & 1:
        spec["wildcards"].append(["N", int(v[6] % 5) - 2])
    if v[7] & 1:
        spec["mutations"] += [["a", "g", int(v[7] % 7) - 3], ["g", "a", int(v[8] % 7) - 3], ["T", "c", 2]]
    sc = S.make_scoring(spec)
    osc = O.Scoring.from_buffer_copy(bytes(sc))
    max_len = int(2 + v[10] % args.max_len) if v[12] % 3 else int(2 + v[10] % 120)
    n = 6 if max_len > 400 else 16
    batch = W.ragged(n, seed=int(v[9]), max_len=max_len, lower_frac=0.25, extra=b"N" if spec["wildcards"] else b"")
    # keep the oracle affordable: long a against short b
    pairs = [(batch.seq_a(p), batch.seq_b(p)[:max(1, 60000 // (len(batch.seq_a(p)) + 1))]) for p in range(batch.n_pairs)]
    batch = W.from_pairs(pairs)
    for is_sw in (0, 1):
        if not is_sw and min(osc.gap_open + osc.gap_extend, osc.gap_extend) < -abs(osc.min_penalty):
            continue                               # outside the parity domain
        h = ctx.upload_scoring(sc, is_sw)
        want = [O.oracle_fill(osc, a, b, is_sw) for a, b in pairs]
        for kid in KERNELS:
            db = S.DeviceBatch(batch, 0, pad_cells=1 if v[13] & 1 else 32, placement="packed")
            db.M.fill_(-3); db.A.fill_(-3); db.B.fill_(-3)
            db.fill(ctx, h, kid)
            torch.cuda.synchronize()
            for p, (rc, M, A, B) in enumerate(want):
                assert rc == 0
                gM, gA, gB = db.pair_matrices(p)
                if not (np.array_equal(gM, M) and np.array_equal(gA, A) and np.array_equal(gB, B)):
                    print("MISMATCH", S.KERNEL_NAMES[kid], "sw" if is_sw else "nw", spec, "pair", p, pairs[p], flush=True)
                    sys.exit(1)
            pairs_checked += len(want)
        ctx.release_scoring(h)
    trials += 1
print(f"fuzz ok: {trials} random scorings x batches, {pairs_checked} pair x kernel checks, max_len {args.max_len}, seed {args.seed}")
