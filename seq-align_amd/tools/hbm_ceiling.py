#!/usr/bin/env python3
"""Measured HBM write / copy ceilings of this MI355X (torch fill_ and copy_ on a
buffer the size of the C2 output, 2.74 GB).  Context for roofline.frac: pure
write streams do not reach the 8 TB/s datasheet figure."""
import json
import sys

import torch

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_739_120_000 // 4
x = torch.empty(n, dtype=torch.int32, device="cuda")
y = torch.empty(n, dtype=torch.int32, device="cuda")
out = {}
for name, fn, bytes_moved in (("fill_write", lambda: x.fill_(7), 4 * n),
                              ("copy_read_write", lambda: y.copy_(x), 8 * n)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(22)]
    for i in range(21):
        ev[i].record(); fn()
    ev[21].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(21))[10]
    out[name] = {"ms": ms, "GBps": bytes_moved / ms / 1e6}
print(json.dumps(out))
