#!/usr/bin/env python3
"""gap_extend > 0 on a C2-shaped batch (10 k NW pairs, 150 x 150): every fill kernel's time and fraction of the
8 TB/s roofline, next to the same batch with the default gap_extend = -1.  Until round 3 every positive extension
was routed to the anti-diagonal kernel; the row sweeps now take the trend of their gap_b scan from the right end."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT / "tests"))
import seqalign_amd as S  # noqa: E402
import orclib as O  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402

out = {}
with S.Context(0) as ctx:
    batch = W.dna_nw_150(10000, seed=1)
    db = S.DeviceBatch(batch, 0, ctx=ctx)
    alg = db.algorithmic_bytes()
    out["arena_placement_quality"] = round(db.placement_quality, 3)
    for ext in (-1, 1, 2):
        sc = S.make_scoring({"init": [1, -2, -4, ext, 0, 0, 0, 0, 0, 0]})
        osc = O.Scoring.from_buffer_copy(bytes(sc))
        h = ctx.upload_scoring(sc, 0)
        row = {}
        for name, k in (("auto", S.KERNEL_AUTO), ("stream", S.KERNEL_STREAM), ("rowscan", S.KERNEL_ROWSCAN), ("wavefront", S.KERNEL_WAVEFRONT)):
            ms = float(np.median(db.time_fill_ms(ctx, h, k, 12)[2:]))
            row[name] = {"ms": round(ms, 4), "frac_of_8TBs": round(alg / (ms * 1e-3) / 8e12, 3)}
        db.fill(ctx, h, S.KERNEL_AUTO)
        db.torch.cuda.synchronize()
        ok = True
        for p in range(0, batch.n_pairs, 997):
            rc, M, A, B = O.oracle_fill(osc, batch.seq_a(p), batch.seq_b(p), 0)
            gM, gA, gB = db.pair_matrices(p)
            ok &= rc == 0 and np.array_equal(M, gM) and np.array_equal(A, gA) and np.array_equal(B, gB)
        row["bit_exact_vs_oracle"] = bool(ok)
        out[f"gap_extend={ext:+d}"] = row
        ctx.release_scoring(h)
print(json.dumps(out, indent=1))
