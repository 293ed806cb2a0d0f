import sys, time
sys.path.insert(0, "seq-align_amd/python"); sys.path.insert(0, ".")
import torch, numpy as np
import seqalign_amd as S
from seqalign_amd import workloads as W
from bench import WORKLOADS
ctx = S.Context(0)
gen, kwargs, _, is_sw, spec, _ = WORKLOADS["C2"]
sc = S.make_scoring(spec)
for n in (256, 512, 1024, 2048, 4096, 6144, 8192):
    batch = getattr(W, gen)(n, **kwargs)
    out = []
    for pk in (0, 1, 0, 1):
        ctx.set_option("pack16", pk)
        ts = []
        for it in range(12):
            t0 = time.perf_counter(); ctx.nw_batch(batch, sc, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
        out.append("%d:%.3f" % (pk, float(np.median(ts[4:]))))
    print("NW n", n, " ".join(out), flush=True)
gen, kwargs, _, is_sw, spec, _ = WORKLOADS["C3"]
sc = S.make_scoring(spec)
for n in (256, 1024, 2048, 4096, 8192):
    batch = getattr(W, gen)(n, **kwargs)
    thr = W.default_minscore(sc.match, int(batch.len_a[0]), int(batch.len_b[0]))
    out = []
    for pk in (0, 1, 0, 1):
        ctx.set_option("pack16", pk)
        ts = []
        for it in range(8):
            t0 = time.perf_counter(); ctx.sw_batch(batch, sc, thr, max_hits=4, hit_cap=4 * n + 8, raw=True); ts.append((time.perf_counter() - t0) * 1e3)
        out.append("%d:%.3f" % (pk, float(np.median(ts[3:]))))
    print("SW n", n, " ".join(out), flush=True)
