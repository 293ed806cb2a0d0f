#!/usr/bin/env python3
"""Does device memory come back after seqalign_arenas_free?  (the steps of tests/test_gpu_parity.py::test_arena_allocator, with the free
memory printed after each)"""
import sys, time, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
import torch, seqalign_amd as S
lib = S.lib()
ctx = S.Context(0)
def free(): return torch.cuda.mem_get_info(0)[0] / 2**20
print("start %.0f MiB" % free())
for nbytes, placed in ((1 << 20, False), (300 << 20, True), ((1 << 30) + 12345 * 4096, True)):
    ptrs = (C.c_void_p * 3)(); q = C.c_float(0)
    assert lib.seqalign_arenas_alloc(ctx._h, C.c_uint64(nbytes), ptrs, C.byref(q)) == 0
    print(nbytes >> 20, "MiB allocated: free %.0f" % free())
    if placed:
        n = nbytes // 4
        for k, p_ in enumerate(ptrs):
            t = torch.as_tensor(S._RawDeviceInts(p_, n), device="cuda:0")
            t.fill_(k + 1)
            idx = torch.tensor([0, n - 1], device="cuda:0")
            assert int(t.sum().item()) == (k + 1) * n and bool((t[idx] == k + 1).all())
            del t
        print("  after the torch checks: free %.0f, torch reserved %.0f" % (free(), torch.cuda.memory_reserved(0) / 2**20))
    assert lib.seqalign_arenas_free(ctx._h, ptrs) == 0
    torch.cuda.synchronize()
    print("  after free: %.0f" % free())
torch.cuda.empty_cache()
print("after torch.cuda.empty_cache(): %.0f" % free())
