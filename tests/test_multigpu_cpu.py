"""The N>1 path on CPU: world_size-2 `gloo` run of the same sharding + reduction
plumbing bench.py uses on RCCL (seqalign_amd.dist, workloads.shard_range).

No GPU here, so each rank's "fill" is the oracle (test infrastructure) -- what is
under test is that contiguous pair-index shards with NO data-path collective
reproduce the single-process result, and that the timing/cell reductions are
MAX / SUM over ranks.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np

import orclib as O
from seqalign_amd import workloads as W

ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "seq-align_amd", "python"))
import numpy as np
import orclib as O
from seqalign_amd import workloads as W
from seqalign_amd.dist import Group

grp = Group("gloo")
full = W.dna_nw_150(37, seed=9, length=40)            # 37: not divisible by 2
mine = full.shard(grp.rank, grp.world)
lo, hi = W.shard_range(full.n_pairs, grp.rank, grp.world)
sc = O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
digests = []
for p in range(mine.n_pairs):
    rc, M, A, B = O.oracle_fill(sc, mine.seq_a(p), mine.seq_b(p), 0)
    digests.append([lo + p, O.fnv(M), O.fnv(A), O.fnv(B)])
grp.barrier()
elapsed = grp.max_float(1.0 + grp.rank)               # MAX over ranks -> world
cells = grp.sum_int(mine.cells())
kernel = grp.broadcast_int(3 if grp.rank == 0 else 99, 0)
gathered = grp.gather_objects(digests)
if grp.rank == 0:
    print("RESULT " + json.dumps(dict(elapsed=elapsed, cells=cells, kernel=kernel, world=grp.world,
                                      digests=[d for part in gathered for d in part])))
grp.close()
'''


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 7, 10000, 1000003):
        for world in (1, 2, 3, 8):
            edges = [W.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_gloo_reproduce_single_process():
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write("ROOT=%r\n" % str(ROOT) + WORKER)
        path = f.name
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), path]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    finally:
        os.unlink(path)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0]
    res = json.loads(line[len("RESULT "):])

    full = W.dna_nw_150(37, seed=9, length=40)
    sc = O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
    want = []
    for p in range(full.n_pairs):
        rc, M, A, B = O.oracle_fill(sc, full.seq_a(p), full.seq_b(p), 0)
        want.append([p, O.fnv(M), O.fnv(A), O.fnv(B)])
    assert res["world"] == 2
    assert res["digests"] == want                  # shards concatenate to the whole, in order
    assert res["cells"] == full.cells()            # SUM over ranks
    assert res["elapsed"] == 2.0                   # MAX over ranks
    assert res["kernel"] == 3                      # broadcast from rank 0
