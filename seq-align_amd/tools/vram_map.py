#!/usr/bin/env python3
"""Map how the fill kernel's speed depends on the physical distance between its
output arenas: M and A are allocated first, then 8 GB balloons are allocated one
after another and after each a new candidate B arena; the stream kernel is timed
on (M, A, B_j), (M, B_j-1, B_j) and (B_j-2, B_j-1, B_j).

Round-1 finding on MI355X (profiles/r01_vram_map.txt): slow (0.52 ms on C2) when
the arenas are within ~16 GB of each other, fast (0.40-0.42 ms) from ~24 GB on,
slow again around 128 GiB: the relation is periodic in physical distance."""
import sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python")); sys.path.insert(0, str(ROOT))
import numpy as np, torch
import seqalign_amd as S
from seqalign_amd import workloads as W
from bench import WORKLOADS
gen, kwargs, per_gpu, is_sw, spec, _ = WORKLOADS["C2"]
ctx = S.Context(0); h = ctx.upload_scoring(S.make_scoring(spec), is_sw)
batch = getattr(W, gen)(per_gpu, **kwargs)
db = S.DeviceBatch(batch, 0, placement="packed")
print("single alloc: %.4f" % float(np.median(db.time_fill_ms(ctx, h, S.KERNEL_STREAM, 20))))
cells = db.total_cells
def arena():
    return torch.empty(cells + 1024, dtype=torch.int32, device="cuda")
def t3(m, a, b):
    db.M, db.A, db.B = m[:cells], a[:cells], b[:cells]
    db.desc.match_scores, db.desc.gap_a_scores, db.desc.gap_b_scores = m.data_ptr(), a.data_ptr(), b.data_ptr()
    return float(np.median(db.time_fill_ms(ctx, h, S.KERNEL_STREAM, 8)))
M, A = arena(), arena()
cands = [arena()]
print("depth GB, (M,A,Bj) us, (M,Bj-1,Bj) us, (Bj-2,Bj-1,Bj) us")
balloons = []
free, total = torch.cuda.mem_get_info()
print("free %.1f GB of %.1f" % (free / 1e9, total / 1e9))
depth = 0
while True:
    free, _ = torch.cuda.mem_get_info()
    if free < 14e9:
        break
    balloons.append(torch.empty(int(8e9), dtype=torch.uint8, device="cuda"))
    depth += 8
    cands.append(arena())
    a = t3(M, A, cands[-1])
    b = t3(M, cands[-2], cands[-1])
    c = t3(cands[-3], cands[-2], cands[-1]) if len(cands) >= 3 else 0
    print(depth, round(1000 * a), round(1000 * b), round(1000 * c), flush=True)
