#!/usr/bin/env python3
"""Stage breakdown of seqalign_sw_batch(max_hits = 4) (option timing=1: laps on stderr) on C3 and C4."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import seqalign_amd as S  # noqa: E402
from seqalign_amd import workloads as W  # noqa: E402
from bench import WORKLOADS  # noqa: E402

hits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
with S.Context(0) as ctx:
    for name in ("C3", "C4"):
        gen, kwargs, n, is_sw, spec, _ = WORKLOADS[name]
        sc = S.make_scoring(spec)
        batch = getattr(W, gen)(n, **kwargs)
        thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
        ctx.sw_batch(batch, sc, thr, max_hits=hits, hit_cap=hits * n + 8, raw=True)
        ctx.sw_batch(batch, sc, thr, max_hits=hits, hit_cap=hits * n + 8, raw=True)
        ctx.set_option("timing", 1)
        for rep in range(3):
            sys.stderr.write(f"---- {name} call {rep}\n")
            t0 = time.perf_counter()
            nh = ctx.sw_batch(batch, sc, thr, max_hits=hits, hit_cap=hits * n + 8, raw=True)[0]
            sys.stderr.write(f"---- {name} wall {(time.perf_counter() - t0) * 1e3:.3f} ms, {nh} hits\n")
        ctx.set_option("timing", 0)
