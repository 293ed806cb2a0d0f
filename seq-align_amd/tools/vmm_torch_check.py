#!/usr/bin/env python3
"""Which torch operations work on arenas built with hipMemCreate / hipMemMap (seqalign_arenas_alloc)?"""
import ctypes as C
import faulthandler
import sys
from pathlib import Path

faulthandler.enable()
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
import torch  # noqa: E402
import seqalign_amd as S  # noqa: E402

lib = S.lib()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 912642048
with S.Context(0) as ctx:
    ptrs = (C.c_void_p * 3)()
    q = C.c_float()
    assert lib.seqalign_arenas_alloc(ctx._h, C.c_uint64(size), ptrs, C.byref(q)) == 0
    print("quality", q.value, [hex(p) for p in ptrs], flush=True)
    n = size // 4
    t = torch.as_tensor(S._RawDeviceInts(ptrs[0], n), device="cuda:0")
    u = torch.as_tensor(S._RawDeviceInts(ptrs[1], n), device="cuda:0")
    steps = [
        ("fill_", lambda: t.fill_(3)),
        ("sum int32", lambda: t.sum().item()),
        ("slice to int64 sum (1M)", lambda: t[:1 << 20].to(torch.int64).sum().item()),
        ("slice to int64 sum (128M + 5)", lambda: t[: (128 << 20) + 5].to(torch.int64).sum().item()),
        ("slice across the chunk seam to int64", lambda: t[(128 << 20) - 1000:(128 << 20) + 1000].to(torch.int64).sum().item()),
        ("whole to int64 sum", lambda: t.to(torch.int64).sum().item()),
        ("clone", lambda: t.clone().sum().item()),
        ("copy_ between arenas", lambda: u.copy_(t).sum().item()),
        ("equal", lambda: torch.equal(t, u)),
        ("cpu slice", lambda: t[5:50].cpu().sum().item()),
        ("cpu whole", lambda: t.cpu().sum().item()),
    ]
    for name, fn in steps:
        print(name, "...", end=" ", flush=True)
        print(fn(), flush=True)
        torch.cuda.synchronize()
    del t, u
    assert lib.seqalign_arenas_free(ctx._h, ptrs) == 0
print("done")
