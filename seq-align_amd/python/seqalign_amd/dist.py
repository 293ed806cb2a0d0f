"""Multi-GPU plumbing for the pair-sharded layout.

The data path has NO collective: rank g owns pairs [g*n/G, (g+1)*n/G)
(workloads.shard_range) and fills them on its own device.  What the ranks exchange
is control plane only: (a) lining up around the timed region and (b) combining the
per-rank timings / counters -- a handful of scalars per run.

Backends:
  "store" (default)  a key-value store (torch.distributed.TCPStore) at MASTER_ADDR:MASTER_PORT --
                     the one torch.distributed.run already hosts when we run under it, our own
                     (rank 0) otherwise.  Every operation is "each rank posts its value, each rank
                     reads all of them": no process group, no collective library, nothing printed.
  "gloo" / "nccl"    torch.distributed process groups ("nccl" IS RCCL on ROCm) doing the same with
                     all_reduce / all_gather_object.  Their C++ start-up banners go to stdout; the
                     group is created with stdout pointed at stderr so that rank 0's stdout stays
                     the ONE JSON line bench.py promises.
"""
from __future__ import annotations

import os
import pickle
import sys
from contextlib import contextmanager
from datetime import timedelta


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


@contextmanager
def stdout_to_stderr():
    """File-descriptor level: C++ libraries (gloo's "[Gloo] Rank 0 is connected to ..." banner) write to fd 1."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class Group:
    """Thin wrapper so that world_size 1 needs no communication at all."""
    _made = 0   # store-backed Groups this process has created (part of their key prefix)

    def __init__(self, backend: str | None = None, device=None):
        self.rank, self.local_rank, self.world = env_world()
        self.device = device
        self.backend = backend or "store"
        self.dist = None
        self.store = None
        self._seq = 0
        if self.world <= 1:
            return
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if self.backend == "store":
            from torch.distributed import PrefixStore, TCPStore
            # under torch.distributed.run the agent already serves a store on MASTER_PORT: be a client of it
            agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
            with stdout_to_stderr():
                tcp = TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), self.world,
                               is_master=(self.rank == 0 and not agent), timeout=timedelta(seconds=1800),
                               wait_for_workers=False)
            self._tcp = tcp
            # The agent's store outlives worker restarts, and a process may make several Groups: every incarnation gets
            # its own key space -- the restart count torchrun hands the workers, and the number of Groups this process has
            # made before (every rank makes them in the same order) -- or a barrier would return at once on the keys of
            # the previous incarnation and max / sum / gather would read its values.
            incarnation = f"r{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}/g{Group._made}"
            Group._made += 1
            self.store = PrefixStore(f"seqalign/{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}/{incarnation}/", tcp)
        else:
            import torch.distributed as dist
            kwargs = {}
            if self.backend == "nccl" and device is not None:
                kwargs["device_id"] = device
            with stdout_to_stderr():
                dist.init_process_group(self.backend, **kwargs)
                dist.barrier()          # the transports connect (and announce themselves) here, not at the first timed barrier
            self.dist = dist

    # ---- the one primitive: everybody's value, in rank order -------------------------------------------------
    def _all(self, value):
        if self.store is not None:
            self._seq += 1
            self.store.set(f"{self._seq}/{self.rank}", pickle.dumps(value))
            keys = [f"{self._seq}/{r}" for r in range(self.world)]
            self.store.wait(keys)
            return [pickle.loads(self.store.get(k)) for k in keys]
        out = [None] * self.world
        self.dist.all_gather_object(out, value)
        return out

    def _tensor(self, value, dtype):
        import torch
        return torch.tensor([value], dtype=dtype, device=self.device if self.device is not None else "cpu")

    def barrier(self):
        if self.store is not None:
            self._all(None)
        elif self.dist:
            self.dist.barrier()

    def max_float(self, x: float) -> float:
        if self.store is not None:
            return float(max(self._all(float(x))))
        if not self.dist:
            return x
        import torch
        t = self._tensor(x, torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, x: int) -> int:
        if self.store is not None:
            return int(sum(self._all(int(x))))
        if not self.dist:
            return x
        import torch
        t = self._tensor(x, torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def broadcast_int(self, x: int, src: int = 0) -> int:
        if self.store is not None:
            return int(self._all(int(x))[src])
        if not self.dist:
            return x
        import torch
        t = self._tensor(x, torch.int64)
        self.dist.broadcast(t, src)
        return int(t.item())

    def gather_objects(self, obj):
        """All ranks' objects, in rank order (reporting: per-rank kernel times, placement, digests)."""
        if self.store is None and not self.dist:
            return [obj]
        return self._all(obj)

    def close(self):
        if self.store is not None:
            # rank 0 may be serving the store: it leaves last
            self.store.add("closing", 1)
            if self.rank == 0:
                import time
                deadline = time.time() + 60
                while self.store.add("closing", 0) < self.world and time.time() < deadline:
                    time.sleep(0.01)
            self.store = None
            self._tcp = None
        if self.dist:
            self.dist.destroy_process_group()
            self.dist = None
