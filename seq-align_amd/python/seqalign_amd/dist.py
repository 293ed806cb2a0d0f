"""Multi-GPU plumbing for the pair-sharded layout.

The data path has NO collective: rank g owns pairs [g*n/G, (g+1)*n/G)
(workloads.shard_range) and fills them on its own device.  torch.distributed is
used only to (a) line ranks up around the timed region and (b) combine the
per-rank timings / counters.  Backend "nccl" is RCCL on ROCm; the CPU tests run
the same code over "gloo".
"""
from __future__ import annotations

import os


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


class Group:
    """Thin wrapper so that world_size 1 needs no process group at all."""

    def __init__(self, backend: str | None = None, device=None):
        self.rank, self.local_rank, self.world = env_world()
        self.device = device
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kwargs = {}
            if backend == "nccl" and device is not None:
                kwargs["device_id"] = device
            dist.init_process_group(backend or "gloo", **kwargs)
            self.dist = dist

    def _tensor(self, value, dtype):
        import torch
        return torch.tensor([value], dtype=dtype, device=self.device if self.device is not None else "cpu")

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_float(self, x: float) -> float:
        if not self.dist:
            return x
        import torch
        t = self._tensor(x, torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, x: int) -> int:
        if not self.dist:
            return x
        import torch
        t = self._tensor(x, torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def broadcast_int(self, x: int, src: int = 0) -> int:
        if not self.dist:
            return x
        import torch
        t = self._tensor(x, torch.int64)
        self.dist.broadcast(t, src)
        return int(t.item())

    def gather_objects(self, obj):
        """All ranks' objects, in rank order (test / reporting use only)."""
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()
            self.dist = None
