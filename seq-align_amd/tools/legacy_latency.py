#!/usr/bin/env python3
"""Latency of the reference-shaped single-pair API (needleman_wunsch_align /
smith_waterman_align + fetch): one H2D + launch + D2H per call."""
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402

lib = S.lib()
out = {}
for name, la, lb in (("9x10", 9, 10), ("150x150", 150, 150), ("150x1000", 150, 1000)):
    batch = W.dna_sw_read_vs_ref(64, seed=3, read_len=la, ref_len=lb) if lb > la + 16 else W.ragged(64, seed=3, max_len=la) if la < 20 else W.dna_nw_150(64, seed=3, length=la)
    sc = S.make_scoring({"preset": "default"})
    nw = C.c_void_p(lib.needleman_wunsch_new())
    res = C.c_void_p(lib.alignment_create(C.c_size_t(256)))
    pairs = [(batch.seq_a(p), batch.seq_b(p)) for p in range(64)]
    for a, b in pairs[:4]:
        lib.needleman_wunsch_align2(a, b, C.c_size_t(len(a)), C.c_size_t(len(b)), C.byref(sc), nw, res)
    t0 = time.perf_counter()
    for a, b in pairs:
        lib.needleman_wunsch_align2(a, b, C.c_size_t(len(a)), C.c_size_t(len(b)), C.byref(sc), nw, res)
    dt = (time.perf_counter() - t0) / len(pairs)
    out[f"needleman_wunsch_align2_{name}"] = dict(us_per_call=dt * 1e6, gcups=la * lb / dt / 1e9)
    lib.alignment_free(res)
    lib.needleman_wunsch_free(nw)
print(json.dumps(out, indent=1))
