#!/usr/bin/env python3
"""Ragged batches through the packed two-pairs-per-wave fills (bucketing by shape):
    NW: 125 000 pairs with len_a, len_b uniform in 100..150  vs  125 000 pairs of 150 x 150 (and of 125 x 125: the same cells)
    SW: C3 with the reads trimmed to 120..150                vs  C3
seqalign_nw_batch / seqalign_sw_batch wall clock (median of 7 after 3), pack16 = 1 and 0, results compared between the two and
(a sample) with the oracle; which kernels ran (seqalign_ctx_last_call_info)."""
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402

import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402
import orclib as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
rng = W.Rng(2024)
dna = np.frombuffer(b"ACGT", np.uint8)


def ragged_nw(n, lo, hi):
    a = dna[rng.below(4, n * hi).astype(np.int64)].reshape(n, hi)
    b = dna[rng.below(4, n * hi).astype(np.int64)].reshape(n, hi)
    keep = rng.unit(n * hi).reshape(n, hi) < 0.9
    b = np.where(keep, a, b)                       # related pairs: alignments with long runs
    la = lo + rng.below(hi - lo + 1, n).astype(np.int64)
    lb = lo + rng.below(hi - lo + 1, n).astype(np.int64)
    return W.from_pairs([(a[k, :la[k]].tobytes(), b[k, :lb[k]].tobytes()) for k in range(n)])


def timed(fn, reps=10):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts[3:]), out


sc = S.make_scoring({"preset": "default"})
osc = O.Scoring.from_buffer_copy(bytes(sc))
with S.Context(0) as ctx:
    batches = [("150 x 150", W.dna_nw_indexed(0, n, seed=5)), ("125 x 125", W.dna_nw_indexed(0, n, seed=5, length=125)),
               ("100..150 x 100..150", ragged_nw(n, 100, 150))]
    base = None
    for name, batch in batches:
        row = {}
        for pk in (1, 0):
            ctx.set_option("pack16", pk)
            ms, out = timed(lambda: ctx.nw_batch(batch, sc, raw=True))
            row[pk] = (ms, [np.array(x, copy=True) for x in out[1:]], ctx.last_call())
        same = all(np.array_equal(x, y) for x, y in zip(row[1][1], row[0][1]))
        ok = True
        str_off, out_a, out_b, out_len, out_score = ctx.nw_batch(batch, sc, raw=True)
        for p in range(0, n, max(1, n // 200)):
            rc, s_, ra, rb = O.oracle_nw(osc, batch.seq_a(p), batch.seq_b(p))
            o, ln = int(str_off[p]), int(out_len[p])
            ok &= rc == 0 and (int(out_score[p]), out_a[o:o + ln].tobytes(), out_b[o:o + ln].tobytes()) == (s_, ra, rb)
        cells = float(batch.cells())
        print(f"NW {n} pairs {name:22s} pack16=1 {row[1][0]:7.3f} ms ({cells / row[1][0] / 1e6:6.1f} GCUPS)  pack16=0 {row[0][0]:7.3f} ms  "
              f"{'identical' if same else 'DIFFERENT'}  oracle sample {'ok' if ok else 'MISMATCH'}  {row[1][2]}", flush=True)
    ctx.set_option("pack16", 1)

    # ---- SW: C3 (10 000 reads of 150 against windows of 1 000) with the reads trimmed to 120..150
    from bench import WORKLOADS
    gen, kwargs, n3, _, spec, _ = WORKLOADS["C3"]
    c3 = getattr(W, gen)(n3, **kwargs)
    cut = 120 + rng.below(31, n3).astype(np.int64)
    trimmed = W.from_pairs([(c3.seq_a(p)[: int(cut[p])], c3.seq_b(p)) for p in range(n3)])
    sw = S.make_scoring(spec)
    osw = O.Scoring.from_buffer_copy(bytes(sw))
    thr = W.default_minscore(sw.match, 150, 1000)
    for name, batch in (("C3", c3), ("C3, reads trimmed to 120..150", trimmed)):
        for max_hits in (1, 4):
            row = {}
            for pk in (1, 0):
                ctx.set_option("pack16", pk)
                ms, out = timed(lambda: ctx.sw_batch(batch, sw, thr, max_hits=max_hits, hit_cap=max_hits * n3 + 8, raw=True))
                import ctypes as C
                nh = int(out[0])
                used = (out[1][nh - 1].str_off + out[1][nh - 1].length + 1) if nh else 0
                row[pk] = (ms, (nh, C.string_at(out[1], nh * C.sizeof(S.SwHit)), bytes(out[2][:used]), bytes(out[3][:used])), ctx.last_call())
            ctx.set_option("pack16", 1)
            got = ctx.sw_batch(batch, sw, thr, max_hits=max_hits, hit_cap=max_hits * n3 + 8)
            ok = True
            for p in range(0, n3, 97):
                rc, want = O.oracle_sw(osw, batch.seq_a(p), batch.seq_b(p), thr, max_hits)
                ok &= rc == 0 and got[p] == want
            print(f"SW {name:32s} max_hits {max_hits}  pack16=1 {row[1][0]:7.3f} ms  pack16=0 {row[0][0]:7.3f} ms  "
                  f"{'identical' if row[1][1] == row[0][1] else 'DIFFERENT'}  oracle sample {'ok' if ok else 'MISMATCH'}  {row[1][2]}", flush=True)
