#!/usr/bin/env python3
"""Differential fuzz of the end-to-end calls (device traceback, device multi-hit enumeration)
against the oracle: seqalign_nw_batch scores + strings, seqalign_sw_batch hit lists.

    python seq-align_amd/tools/fuzz_e2e.py --seconds 300

tests/test_gpu_soak.py runs a seeded slice of it (run(seconds, seed, max_trials)) under the `gpu` marker.
"""
import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import orclib as O  # noqa: E402
import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402



def run(seconds=120.0, seed=1, max_trials=1 << 60, ctx=None):
    """Fuzz until `seconds` have passed or `max_trials` scorings were drawn; SystemExit(1) on the first mismatch."""
    rng = W.Rng(seed)
    ctx = ctx or S.Context(0)
    t_end = time.time() + seconds
    trials = nw_checked = sw_checked = 0

    def rand(n, alpha=b"ACGT"):
        return bytes(alpha[i] for i in rng.below(len(alpha), n)) if n else b""

    while time.time() < t_end and trials < max_trials:
        v = rng.below(1 << 20, 16).astype(int)
        flags = [int(v[0] >> k) & 1 for k in range(5)]
        if v[15] % 5 < 3:
            flags = [0, 0, 0, 0, 0]                   # plain scorings (the direction-byte fills' domain): 3 draws in 5
        match, mismatch = int(1 + v[1] % 4), -int(v[2] % 5)
        go, ge = -int(v[3] % 8), -int(v[4] % 3)
        if flags[2] and flags[3]:
            mismatch = min(mismatch, go + ge)
        spec = {"init": [match, mismatch, go, ge, *flags, int(v[5] & 1)], "wildcards": [["N", int(v[6] % 3) - 1]] if v[6] & 1 else []}
        sc = S.make_scoring(spec)
        osc = O.Scoring.from_buffer_copy(bytes(sc))
        # pairs: random, related, and tandem repeats (many equal-score candidates, several hits)
        pairs = []
        mode = int(v[0] >> 9) % 8                     # (which sweep form: below)
        lb_max = 1500 if mode == 6 else 200           # mode 6: long second sequences too, so that the sweep's word is 32 bits of
                                                      # fields, 32 bits in mixed radix or 64 bits from draw to draw
        for k in range(12):
            kind = int(v[7] + k) % 3
            la = int(2 + (v[8] * (k + 1)) % (260 if k % 4 else 900))
            if kind == 0:
                a, b = rand(la), rand(int(2 + (v[9] * (k + 3)) % lb_max))
            elif kind == 1:
                a = rand(la)
                cut = int(v[10] % max(1, len(a)))
                b = rand(int(v[11] % 30)) + a[cut:cut + 120] + rand(int(v[12] % 30))
            else:
                unit = rand(int(2 + v[13] % 7))
                a, b = unit * int(2 + v[14] % 20), rand(3) + unit * int(2 + v[15] % 25)
            if spec["wildcards"] and k % 5 == 0:
                a = a[:len(a) // 2] + b"N" + a[len(a) // 2:]
            pairs.append((a, b))
        batch = W.from_pairs(pairs)
        ctx.set_option("trace_kernel", ("lane", "wave")[int(v[0] >> 7) & 1])
        ctx.set_option("walk_group", (0, 1, 4, 8)[int(v[0] >> 17) & 3])   # the tile walker: one walk per wave, or 4 / 8 in lockstep
        ctx.set_option("walk_tile", (0, 32, 64, 0)[int(v[0] >> 21) & 3])      # the local tile walker's tile edge
        ctx.set_option("dirs_local", 0 if (int(v[0] >> 19) & 3) == 0 else 1)   # the direction byte's local form (tile walks) / a quarter of the draws: the older form
        # multi-hit enumeration (the reverse sweep): segments of 64 / 128 / 256 columns with the winners of two rows in
        # LDS, one wave per 256-column strip, or behind a fill that cannot report the candidates' box and rows
        for key in ("sweep_cpl", "sweep_mode", "kernel", "subbatches", "nw_dirs", "sweep_dirs", "pack16"):
            ctx.set_option(key, S.OPTION_DEFAULTS[key])
        if mode < 3:
            ctx.set_option("sweep_cpl", (1, 2, 4)[mode])
        elif mode == 3:
            ctx.set_option("sweep_mode", "strips")
        elif mode == 4:
            ctx.set_option("kernel", "rowscan")
        elif mode == 5:
            ctx.set_option("kernel", "wgstream")   # (rows over 512 columns: reports the candidates itself; else falls back)
        elif mode == 6:
            # one wave per pair whatever the batch's size: rows of 513 .. 1 024 columns through the direction bytes and the
            # one-word sweep (12 / 16 columns per lane); every other draw with the packed fills forced on
            ctx.set_option("sweep_mode", "pair")
            if int(v[0] >> 16) & 1: ctx.set_option("pack16", 2)
        ctx.set_option("subbatches", (0, 1, 2, 5)[int(v[0] >> 12) % 4])   # seqalign_nw_batch: pipelined sub-batches
        if int(v[0] >> 14) % 4 == 0:                  # a quarter of the draws through the three-matrix paths whatever the scoring
            ctx.set_option("nw_dirs", 0)
            ctx.set_option("sweep_dirs", 0)
        if min(osc.gap_open + osc.gap_extend, osc.gap_extend) >= -abs(osc.min_penalty):   # NW parity domain
            res = ctx.nw_batch(batch, sc)
            for p, (a, b) in enumerate(pairs):
                rc, s_, ra, rb = O.oracle_nw(osc, a, b)
                if rc != 0 or res[p] != (s_, ra, rb):
                    print("NW MISMATCH", spec, p, pairs[p], res[p], (s_, ra, rb), flush=True)
                    raise SystemExit(1)
            nw_checked += len(pairs)
        thr = int(1 + v[5] % (6 * match))
        max_hits = (1, 2, 5, 16, 40, 1 << 20)[int(v[4] >> 4) % 6]
        got = ctx.sw_batch(batch, sc, thr, max_hits=max_hits, hit_cap=400000)
        for p, (a, b) in enumerate(pairs):
            rc, want = O.oracle_sw(osc, a, b, thr, max_hits)
            if rc != 0 or got[p] != want:
                print("SW MISMATCH", spec, "thr", thr, "max_hits", max_hits, p, pairs[p], flush=True)
                raise SystemExit(1)
        sw_checked += len(pairs)
        trials += 1
    msg = (f"fuzz_e2e ok: {trials} random scorings x batches; {nw_checked} NW alignments, {sw_checked} SW hit lists identical "
           f"to the oracle (seed {seed})")
    print(msg, flush=True)
    for key in ("sweep_cpl", "sweep_mode", "kernel", "subbatches", "nw_dirs", "sweep_dirs", "trace_kernel", "pack16", "walk_group", "dirs_local", "walk_tile"):
        ctx.set_option(key, S.OPTION_DEFAULTS[key])
    return {"trials": trials, "nw_checked": nw_checked, "sw_checked": sw_checked}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--trials", type=int, default=1 << 60)
    args = ap.parse_args()
    run(args.seconds, args.seed, args.trials)
