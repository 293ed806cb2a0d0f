#!/usr/bin/env python3
"""The packed two-pairs-per-wave direction fills (sa_fill_dirs_x2.hip, option pack16) against the one-pair kernels
(pack16 = 0) and the oracle on uniform batches (pack16 = 2: the packed kernels whatever the batch's size), then their timing on C2 and C5's share.

    python seq-align_amd/tools/x2_check.py [seconds]

tests/test_gpu_soak.py runs a seeded slice of the four checks (check_uniform / check_mixed / check_sw / check_ragged)
under the `gpu` marker.
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import orclib as O  # noqa: E402
import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402

rng = W.Rng(77)
ctx = None
DNA = np.frombuffer(b"ACGT", np.uint8)
DNAN = np.frombuffer(b"ACGTN", np.uint8)
PROT = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVBZX", np.uint8)


shapes = [(1, 1), (1, 7), (7, 1), (3, 5), (63, 64), (64, 64), (64, 10), (127, 130), (128, 128), (150, 150), (151, 149), (191, 40),
          (192, 33), (200, 37), (255, 256), (300, 100), (383, 20), (400, 60), (511, 70)]


def setup(seed=77, context=None):
    """(Re)seed the generator and bind the context the checks run on."""
    global rng, ctx
    rng = W.Rng(seed)
    ctx = context or S.Context(0)
    return ctx


def uniform_batch(n, la, lb, related, DNA=DNA):
    K = len(DNA)
    a = DNA[rng.below(K, n * la).astype(np.int64)].reshape(n, la)
    if related and la == lb:
        b = W._mutate(a, lb, DNA, rng, 0.08, 0.03)
    elif related:
        b = DNA[rng.below(K, n * lb).astype(np.int64)].reshape(n, lb)
        k = min(la, lb)
        b[:, :k] = a[:, :k]
        flip = rng.unit(n * k).reshape(n, k) < 0.1
        b[:, :k] = np.where(flip, DNA[rng.below(K, n * k).astype(np.int64)].reshape(n, k), b[:, :k])
    else:
        b = DNA[rng.below(K, n * lb).astype(np.int64)].reshape(n, lb)
    return W._fixed_batch(a, b)


def same(r0, r1):
    return all(np.array_equal(x, y) for x, y in zip(r0, r1))


def near_bound_spec(la, lb, v, nw=False):
    """A match / mismatch scoring whose int16 admission bound (sa_x2_scores_fit: (la + lb + 2) pen + (la + 1) |ext| <= 30 000; nw: the
    packed NW fills' own, sa_domain_nw_x2_scores_fit: (la + lb + 2) (pen + |ext|) <= 30 000) evaluates to within 2 % of 30 000 for this
    shape -- where the packed halves come closest to leaving int16 -- or None."""
    ge = -int(v[6] % 4)
    if nw:
        pen = 30000 // (la + lb + 2) - abs(ge)
        if pen < 2 or (la + lb + 2) * (pen + abs(ge)) < 29400:
            return None
    else:
        pen = (30000 - (la + 1) * abs(ge)) // (la + lb + 2)
        if pen < 2 or (la + lb + 2) * pen + (la + 1) * abs(ge) < 29400:
            return None
    kind = int(v[5] % 3)          # which penalty sits at the bound: match, mismatch, or the first gap character
    match = pen if kind == 0 else max(1, pen // (2 + int(v[3] % 3)))
    mismatch = -pen if kind == 1 else -max(0, pen // (1 + int(v[4] % 4)))
    open1 = pen if kind == 2 else max(abs(ge), pen // (1 + int(v[5] % 5)))
    go = -(open1 - abs(ge))
    return {"init": [match, mismatch, go, ge, 0, 0, 0, 0, 0, int(v[7] & 1)], "wildcards": []}


def check_uniform(seconds, max_trials=1 << 60):
    """NW on uniform batches: packed fills (pack16 = 2, two / four pairs per wave) vs the one-pair kernels and the oracle."""
    t_end = time.time() + seconds
    trials = pairs = oracle_pairs = quad_batches = 0
    while time.time() < t_end and trials < max_trials:
        v = rng.below(1 << 20, 12).astype(int)
        la, lb = shapes[trials % len(shapes)] if trials < 3 * len(shapes) else (int(1 + v[0] % 511), int(1 + v[1] % 300))
        n = int(1 + v[2] % 700)
        match, mismatch = int(1 + v[3] % 5), -int(v[4] % 6)
        go, ge = -int(v[5] % 12), -int(v[6] % 4)
        spec = {"init": [match, mismatch, go, ge, 0, 0, 0, 0, 0, int(v[7] & 1)], "wildcards": []}
        alpha = DNA
        if trials % 5 == 4 and near_bound_spec(la, lb, v, nw=True):      # a fifth of the scorings at the edge of int16
            spec = near_bound_spec(la, lb, v, nw=True)
            match = spec["init"][0]
        elif v[11] % 5 == 0:      # a substitution table: BLOSUM62 on protein, or a wildcard
            spec, alpha = ({"preset": "BLOSUM62"}, PROT) if v[11] % 2 else ({**spec, "wildcards": [["N", int(v[10] % 3) - 1]]}, DNAN)
        sc = S.make_scoring(spec)
        batch = uniform_batch(n, la, lb, bool(v[8] & 1), alpha)
        ctx.set_option("pack16", 0)
        r0 = [x.copy() for x in ctx.nw_batch(batch, sc, raw=True)]
        ctx._nw_buffers = None
        ctx.set_option("pack16", 2)
        ctx.set_option("quad", 2 if trials % 2 else 1)      # every other batch: four pairs per wave where the shape allows (rows <= 192 columns)
        r1 = [x.copy() for x in ctx.nw_batch(batch, sc, raw=True)]
        quad_batches += "fill_nw_dirs_x4" in ctx.last_call()
        ctx._nw_buffers = None
        if not same(r0, r1):
            bad = np.nonzero(r0[4] != r1[4])[0]
            print("MISMATCH pack16 0 vs 1:", la, lb, n, spec, "score diffs at", bad[:8], flush=True)
            for p in range(n):
                o, l0, l1 = int(r0[0][p]), int(r0[3][p]), int(r1[3][p])
                if l0 != l1 or not np.array_equal(r0[1][o:o + l0], r1[1][o:o + l1]) or not np.array_equal(r0[2][o:o + l0], r1[2][o:o + l1]):
                    print(" pair", p, "score", r0[4][p], r1[4][p], "\n ", r0[1][o:o + l0].tobytes(), "\n ", r0[2][o:o + l0].tobytes(),
                          "\n ", r1[1][o:o + l1].tobytes(), "\n ", r1[2][o:o + l1].tobytes())
                    break
            raise SystemExit(1)
        if trials % 4 == 0:      # the oracle on a few pairs of the batch
            osc = O.Scoring.from_buffer_copy(bytes(sc))
            for p in range(0, n, max(1, n // 6)):
                _, score, sa, sb = O.oracle_nw(osc, batch.seq_a(p), batch.seq_b(p))
                o, ln = int(r1[0][p]), int(r1[3][p])
                if score != int(r1[4][p]) or sa != r1[1][o:o + ln].tobytes() or sb != r1[2][o:o + ln].tobytes():
                    print("MISMATCH vs oracle:", la, lb, n, spec, "pair", p, score, int(r1[4][p]), flush=True)
                    raise SystemExit(1)
                oracle_pairs += 1
        trials += 1
        pairs += n
    print(f"x2_check: {trials} uniform batches ({quad_batches} of them four pairs per wave), {pairs} pairs: packed (pack16 = 2) identical to pack16 = 0; {oracle_pairs} pairs against the oracle", flush=True)
    ctx.set_option("quad", 0)

    return {"batches": trials, "pairs": pairs, "oracle_pairs": oracle_pairs, "quad_batches": quad_batches}


def check_mixed(seconds, max_trials=1 << 60):
    """NW, chunks that are MOSTLY of one shape: the modal shape packed, the others one per wave, in one grid."""
    t_end = time.time() + seconds
    mx_trials = mx_pairs = mx_oracle = 0
    while time.time() < t_end and mx_trials < max_trials:
        v = rng.below(1 << 20, 12).astype(int)
        la, lb = int(1 + v[0] % 400), int(1 + v[1] % 300)
        n = int(4 + v[2] % 400)
        match, mismatch = int(1 + v[3] % 5), -int(v[4] % 6)
        go, ge = -int(v[5] % 12), -int(v[6] % 4)
        spec = {"init": [match, mismatch, go, ge, 0, 0, 0, 0, 0, int(v[7] & 1)], "wildcards": []}
        sc = S.make_scoring(spec)
        odd_every = int(3 + v[8] % 6)
        lens = rng.below(511, 2 * n).astype(int)
        pairs = []
        for k in range(n):
            if k % odd_every == 1:
                xa, xb = int(lens[2 * k]), int(lens[2 * k + 1] % 300)
            else:
                xa, xb = la, lb
            a = bytes(b"ACGT"[i] for i in rng.below(4, xa)) if xa else b""
            b = (a[: xb] + (bytes(b"ACGT"[i] for i in rng.below(4, xb - xa)) if xb > xa else b"")) if k % 2 else (bytes(b"ACGT"[i] for i in rng.below(4, xb)) if xb else b"")
            pairs.append((a, b))
        batch = W.from_pairs(pairs)
        ctx.set_option("subbatches", int(v[9] % 4))
        ctx.set_option("pack16", 0)
        r0 = ctx.nw_batch(batch, sc)
        ctx.set_option("pack16", 2)
        r1 = ctx.nw_batch(batch, sc)
        ctx.set_option("subbatches", 0)
        if r0 != r1:
            bad = [p for p in range(n) if r0[p] != r1[p]]
            print("MIXED MISMATCH pack16 0 vs 2:", la, lb, n, spec, "pairs", bad[:6], flush=True)
            raise SystemExit(1)
        if mx_trials % 4 == 0:
            osc = O.Scoring.from_buffer_copy(bytes(sc))
            for p in range(0, n, max(1, n // 6)):
                _, score, sa, sb = O.oracle_nw(osc, pairs[p][0], pairs[p][1])
                if r1[p] != (score, sa, sb):
                    print("MIXED MISMATCH vs oracle:", la, lb, n, spec, "pair", p, flush=True)
                    raise SystemExit(1)
                mx_oracle += 1
        mx_trials += 1
        mx_pairs += n
    print(f"x2_check: NW mostly-one-shape: {mx_trials} batches, {mx_pairs} pairs: mixed grid (pack16 = 2) identical to pack16 = 0; {mx_oracle} pairs against the oracle", flush=True)

    return {"batches": mx_trials, "pairs": mx_pairs, "oracle_pairs": mx_oracle}


def check_sw(seconds, max_trials=1 << 60):
    """Smith-Waterman multi-hit / best hit: the packed fills vs pack16 = 0 and the oracle."""
    t_end = time.time() + seconds
    sw_trials = sw_pairs = sw_oracle = sw_quad = 0
    while time.time() < t_end and sw_trials < max_trials:
        v = rng.below(1 << 20, 12).astype(int)
        la, lb = shapes[sw_trials % len(shapes)] if sw_trials < 2 * len(shapes) else (int(1 + v[0] % 511), int(1 + v[1] % 300))
        n = int(1 + v[2] % 300)
        match, mismatch = int(1 + v[3] % 5), -int(v[4] % 6)
        go, ge = -int(v[5] % 12), -int(v[6] % 4)
        spec = {"init": [match, mismatch, go, ge, 0, 0, 0, 0, 0, int(v[7] & 1)], "wildcards": []}
        alpha = DNA
        if sw_trials % 5 == 4 and near_bound_spec(la, lb, v):      # a fifth of the scorings at the edge of int16
            spec = near_bound_spec(la, lb, v)
            match = spec["init"][0]
        elif v[11] % 5 == 0:
            spec, alpha = ({"preset": "BLOSUM62"}, PROT) if v[11] % 2 else ({**spec, "wildcards": [["N", int(v[10] % 3) - 1]]}, DNAN)
            if "preset" in spec: match = 5
        sc = S.make_scoring(spec)
        batch = uniform_batch(n, la, lb, bool(v[8] & 1), alpha)
        thr = int(1 + v[9] % max(2, match * min(la, lb) // 2))
        max_hits = int(1 + v[10] % 8) if sw_trials % 3 else 1      # a third: the best hit only (its own fill; every other one four pairs per wave)
        ctx.set_option("pack16", 0)
        r0 = ctx.sw_batch(batch, sc, thr, max_hits=max_hits, hit_cap=8 * n + 8)
        ctx.set_option("pack16", 2)
        ctx.set_option("quad", 2 if sw_trials % 2 else 1)
        r1 = ctx.sw_batch(batch, sc, thr, max_hits=max_hits, hit_cap=8 * n + 8)
        sw_quad += "fill_sw_best_x4" in ctx.last_call()
        if r0 != r1:
            bad = [p for p in range(n) if r0[p] != r1[p]]
            print("SW MISMATCH pack16 0 vs 1:", la, lb, n, spec, "thr", thr, "max_hits", max_hits, "pairs", bad[:6], flush=True)
            print(r0[bad[0]][:3], "\n", r1[bad[0]][:3])
            raise SystemExit(1)
        if sw_trials % 4 == 0:
            osc = O.Scoring.from_buffer_copy(bytes(sc))
            for p in range(0, n, max(1, n // 5)):
                rc, want = O.oracle_sw(osc, batch.seq_a(p), batch.seq_b(p), thr, max_hits)
                if rc != 0 or r1[p] != want:
                    print("SW MISMATCH vs oracle:", la, lb, n, spec, "thr", thr, "max_hits", max_hits, "pair", p, flush=True)
                    raise SystemExit(1)
                sw_oracle += 1
        sw_trials += 1
        sw_pairs += n
    ctx.set_option("quad", 0)
    print(f"x2_check: SW multi-hit / best hit: {sw_trials} uniform batches ({sw_quad} of them the best-hit fill with four pairs per wave), {sw_pairs} pairs: packed (pack16 = 2) hit lists identical to pack16 = 0; {sw_oracle} pairs against the oracle", flush=True)

    return {"batches": sw_trials, "pairs": sw_pairs, "oracle_pairs": sw_oracle, "quad_batches": sw_quad}


def check_ragged(seconds, max_trials=1 << 60):
    """Ragged batches bucketed by shape on the host (NW: mixed grid; SW: pair lists) vs pack16 = 0 and the oracle."""
    t_end = time.time() + seconds
    rg_trials = rg_pairs = rg_oracle = 0
    while time.time() < t_end and rg_trials < max_trials:
        v = rng.below(1 << 20, 12).astype(int)
        la0, lb0 = int(1 + v[0] % 380), int(1 + v[1] % 280)
        da, db = int(1 + v[2] % 12), int(1 + v[3] % 9)          # shapes la0 .. la0 + da - 1  x  lb0 .. lb0 + db - 1
        n = int(4 + v[4] % 500)
        match, mismatch = int(1 + v[5] % 5), -int(v[6] % 6)
        go, ge = -int(v[7] % 12), -int(v[8] % 4)
        spec = {"init": [match, mismatch, go, ge, 0, 0, 0, 0, 0, int(v[9] & 1)], "wildcards": []}
        sc = S.make_scoring(spec)
        lens = rng.below(1 << 16, 2 * n).astype(int)
        pairs = []
        for k in range(n):
            xa, xb = la0 + int(lens[2 * k] % da), lb0 + int(lens[2 * k + 1] % db)
            a = bytes(b"ACGT"[i] for i in rng.below(4, xa))
            b = (a[: xb] + (bytes(b"ACGT"[i] for i in rng.below(4, xb - xa)) if xb > xa else b"")) if k % 2 else bytes(b"ACGT"[i] for i in rng.below(4, xb))
            pairs.append((a, b))
        batch = W.from_pairs(pairs)
        thr = int(1 + v[10] % max(2, match * min(la0, lb0) // 2))
        max_hits = int(1 + v[11] % 6)
        ctx.set_option("subbatches", int(v[9] % 4))
        res = {}
        for pk in (0, 2):
            ctx.set_option("pack16", pk)
            res[pk] = (ctx.nw_batch(batch, sc), ctx.sw_batch(batch, sc, thr, max_hits=max_hits, hit_cap=8 * n + 8))
        ctx.set_option("subbatches", 0)
        if res[0] != res[2]:
            print("RAGGED MISMATCH pack16 0 vs 2:", la0, lb0, da, db, n, spec, "thr", thr, "max_hits", max_hits, flush=True)
            raise SystemExit(1)
        if rg_trials % 4 == 0:
            osc = O.Scoring.from_buffer_copy(bytes(sc))
            for p in range(0, n, max(1, n // 6)):
                _, score, sa, sb = O.oracle_nw(osc, pairs[p][0], pairs[p][1])
                rc, want = O.oracle_sw(osc, pairs[p][0], pairs[p][1], thr, max_hits)
                if res[2][0][p] != (score, sa, sb) or rc != 0 or res[2][1][p] != want:
                    print("RAGGED MISMATCH vs oracle:", la0, lb0, da, db, n, spec, "pair", p, flush=True)
                    raise SystemExit(1)
                rg_oracle += 1
        rg_trials += 1
        rg_pairs += n
    print(f"x2_check: ragged (bucketed by shape): {rg_trials} batches, {rg_pairs} pairs: NW strings and SW hit lists with pack16 = 2 identical to pack16 = 0; {rg_oracle} pairs against the oracle", flush=True)

    ctx.set_option("pack16", S.OPTION_DEFAULTS["pack16"])
    return {"batches": rg_trials, "pairs": rg_pairs, "oracle_pairs": rg_oracle}


def timings():
    from bench import WORKLOADS  # noqa: E402
    for name, n in (("C2", 10000), ("C5share", 125000)):
        gen, kwargs, _, is_sw, spec, _ = WORKLOADS["C2"]
        batch = getattr(W, gen)(n, **kwargs)
        sc = S.make_scoring(spec)
        for pk, wo in ((0, 0), (0, 1), (1, 0), (1, 1), (1, 0), (1, 1)):
            ctx.set_option("pack16", pk)
            ctx.set_option("walk_overlap", wo)
            ts = []
            for it in range(7):
                t0 = time.perf_counter()
                ctx.nw_batch(batch, sc, raw=True)
                ts.append((time.perf_counter() - t0) * 1e3)
            print(f"{name} pack16={pk} walk_overlap={wo}: " + " ".join("%.3f" % t for t in ts[2:]) + " ms", flush=True)

    for name in ("C3",):
        gen, kwargs, n, is_sw, spec, _ = WORKLOADS[name]
        batch = getattr(W, gen)(n, **kwargs)
        sc = S.make_scoring(spec)
        thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
        for pk in (0, 1, 0, 1):
            ctx.set_option("pack16", pk)
            ts = []
            for it in range(6):
                t0 = time.perf_counter()
                nh = ctx.sw_batch(batch, sc, thr, max_hits=4, hit_cap=4 * n + 8, raw=True)[0]
                ts.append((time.perf_counter() - t0) * 1e3)
            print(f"{name} sw_batch(max_hits=4) pack16={pk}: {nh} hits " + " ".join("%.3f" % t for t in ts[2:]) + " ms", flush=True)


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30
    setup(77)
    check_uniform(seconds)
    check_mixed(seconds / 2)
    check_sw(seconds)
    check_ragged(seconds / 2)
    timings()
