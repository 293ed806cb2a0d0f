#!/usr/bin/env python3
"""How much does the fill kernel's time depend on WHERE the three output arenas
live?  Allocates N separate arenas (each big enough for one matrix arena of the
workload) and times the stream kernel for every (M, A, B) triple of them.

    python seq-align_amd/tools/placement_scan.py --workload C2 --arenas 8

Finding (round 1, MI355X): the same kernel on the same data runs 0.46 or 0.53 ms
on C2 depending only on the physical placement of the arenas (stable per
allocation, bimodal; a sequential memset of the same arenas does not care)."""
import argparse
import itertools
import json
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
ap.add_argument("--workload", default="C2")
ap.add_argument("--arenas", type=int, default=8)
ap.add_argument("--launches", type=int, default=8)
ap.add_argument("--linear", action="store_true", help="many arenas: time (0, 1, j) and (j, j+1, j+2) only")
args = ap.parse_args()
gen, kwargs, per_gpu, is_sw, spec, _ = WORKLOADS[args.workload]
batch = getattr(W, gen)(per_gpu, **kwargs)
ctx = S.Context(0)
h = ctx.upload_scoring(S.make_scoring(spec), is_sw)
db = S.DeviceBatch(batch, 0, placement="packed")
base = float(np.median(db.time_fill_ms(ctx, h, S.KERNEL_STREAM, 30)))
print(f"default single-allocation layout: {base:.4f} ms")
cells = db.total_cells
pool = [torch.empty(cells + 1024, dtype=torch.int32, device="cuda") for _ in range(args.arenas)]
for t in pool:
    assert t.data_ptr() % 4096 == 0


def time_triple(i, j, k):
    db.M, db.A, db.B = pool[i][:cells], pool[j][:cells], pool[k][:cells]
    db.desc.match_scores, db.desc.gap_a_scores, db.desc.gap_b_scores = (pool[i].data_ptr(), pool[j].data_ptr(),
                                                                        pool[k].data_ptr())
    return float(np.median(db.time_fill_ms(ctx, h, S.KERNEL_STREAM, args.launches)))


time_triple(0, 1, 2)
if args.linear:
    print("ptr GiB offsets from the lowest:", [round((t.data_ptr() - min(x.data_ptr() for x in pool)) / 2**30, 1) for t in pool])
    print("(0,1,j) us:", [round(1000 * time_triple(0, 1, j)) for j in range(2, args.arenas)])
    print("(j,j+1,j+2) us:", [round(1000 * time_triple(j, j + 1, j + 2)) for j in range(args.arenas - 2)])
    print("(j,j+1,last) us:", [round(1000 * time_triple(j, j + 1, args.arenas - 1)) for j in range(args.arenas - 2)])
    sys.exit(0)
res = {}
for tri in itertools.combinations(range(args.arenas), 3):
    res[tri] = time_triple(*tri)
vals = np.array(sorted(res.values()))
print("triples:", len(vals), " min %.4f  p25 %.4f  median %.4f  p75 %.4f  max %.4f" %
      (vals[0], np.percentile(vals, 25), np.median(vals), np.percentile(vals, 75), vals[-1]))
print("histogram (ms):", np.histogram(vals, bins=8))
best = sorted(res.items(), key=lambda kv: kv[1])[:6]
worst = sorted(res.items(), key=lambda kv: kv[1])[-4:]
print("best:", best)
print("worst:", worst)
# pairwise structure: mean time of the triples containing the pair
n = args.arenas
pm = np.zeros((n, n))
for (i, j, k), v in res.items():
    for a, b in ((i, j), (i, k), (j, k)):
        pm[a, b] += v / (n - 2)
        pm[b, a] += v / (n - 2)
print("mean time of the triples containing pair (i,j), us:")
print(np.round(pm * 1000).astype(int))
print(json.dumps({"workload": args.workload, "single_allocation_ms": base, "triples": len(vals),
                  "min_ms": float(vals[0]), "median_ms": float(np.median(vals)), "max_ms": float(vals[-1])}))
