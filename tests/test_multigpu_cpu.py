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


def test_cell_balanced_shards_on_a_ragged_batch():
    """a batch whose second half is 10x longer is cut at equal CELLS, not equal pair counts;
    blocks stay contiguous (results keep pair order) and cover the batch."""
    cells = np.array([100] * 1000 + [1000] * 1000, dtype=np.uint64)
    for world in (2, 3, 8):
        e = W.shard_edges_cells(cells, world)
        assert e[0] == 0 and e[-1] == 2000 and all(a <= b for a, b in zip(e, e[1:]))
        loads = [int(cells[e[g]:e[g + 1]].sum()) for g in range(world)]
        assert max(loads) - min(loads) <= 1000            # within one (largest) pair
        by_count = [int(cells[W.shard_range(2000, g, world)[0]:W.shard_range(2000, g, world)[1]].sum()) for g in range(world)]
        assert max(loads) < max(by_count)                 # strictly better than splitting by pair count
    # uniform batches: identical to the count-based split
    assert W.shard_edges_cells(np.full(10000, 22801), 8) == [W.shard_range(10000, g, 8)[0] for g in range(8)] + [10000]
    # degenerate: fewer pairs than ranks, empty batch
    assert W.shard_edges_cells([7], 3)[-1] == 1 and W.shard_edges_cells([], 4) == [0, 0, 0, 0, 0]
    b = W.ragged(50, seed=3, max_len=60)
    parts = [b.shard(g, 4) for g in range(4)]
    assert sum(p.n_pairs for p in parts) == 50
    assert b"".join(p.seq_a(i) for p in parts for i in range(p.n_pairs)) == b"".join(b.seq_a(i) for i in range(50))


def _c5_digests(n):
    full = W.dna_nw_indexed(0, n, seed=5)                  # C5's stream: pair p is the same in every shard size
    sc = O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
    want = []
    for p in range(n):
        rc, M, A, B = O.oracle_fill(sc, full.seq_a(p), full.seq_b(p), 0)
        want.append([p, O.fnv(M), O.fnv(A), O.fnv(B)])
    return want, full.cells()


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR",
                                                            "TORCHELASTIC_USE_AGENT_STORE", "SEQALIGN_DIST_BACKEND")}
    env.update(extra)
    return env


def test_bench_launches_its_own_ranks_and_shards_c5():
    """`python bench.py --gpus 8` outside any launcher starts eight ranks itself (no torchrun; the control plane is a
    key-value store, no process group, no RCCL) and shards BASELINE config 5's pair stream by contiguous index.  CPU
    box: --plumbing-test makes the oracle fill a handful of pairs; what is checked is launcher + sharding + MAX/SUM
    reductions -- and that rank 0's stdout is EXACTLY one line (the driver parses it)."""
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--plumbing-test", "--pairs", "3"],
                         env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[:400]      # ONE line, nothing else on stdout
    res = json.loads(lines[0])
    want, cells = _c5_digests(24)
    assert res["n_gpus"] == 8 and res["global_pairs"] == 24 and res["elapsed_max"] == 8.0
    assert res["digests"] == want and res["cells_sum"] == cells


def test_bench_under_torch_distributed_run_prints_one_line():
    """The driver's launch line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N.  The ranks talk through the launcher's own store; stdout of the whole job is the
    one JSON line -- with the store control plane and with a gloo process group (whose C++ banner must go to stderr)."""
    for backend in ("store", "gloo"):
        port = free_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--plumbing-test", "--pairs", "4"]
        out = subprocess.run(cmd, env=_clean_env(SEQALIGN_DIST_BACKEND=backend, OMP_NUM_THREADS="1"), capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, (backend, out.stderr[-2000:])
        lines = [l for l in out.stdout.splitlines() if l.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), (backend, out.stdout[:600])
        res = json.loads(lines[0])
        want, cells = _c5_digests(8)
        assert res["n_gpus"] == 2 and res["digests"] == want and res["cells_sum"] == cells and res["elapsed_max"] == 2.0


def test_rank_pinning_deals_out_a_numa_node():
    """bench.py pins each rank (and so the library's worker pool) to its share of the CPUs of its GPU's NUMA node:
    ranks on the same node get disjoint, equal shares that keep hyper-thread siblings together."""
    sys.path.insert(0, str(ROOT))
    import bench
    assert bench._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    if not hasattr(os, "sched_setaffinity"):
        return
    before = os.sched_getaffinity(0)
    try:
        allowed = sorted(before)
        nodes = [0, 0, 1, 1]                      # ranks 0, 1 on node 0; 2, 3 on node 1
        seen = []
        for rank in (0, 1):
            os.sched_setaffinity(0, before)
            info = bench.pin_rank(0, allowed, rank, 4, nodes)
            mine = sorted(os.sched_getaffinity(0))
            seen.append(mine)
            if len(allowed) >= 2:
                assert info["cpus"] == len(mine) and 1 <= len(mine) <= max(1, len(allowed) // 2) + 1
        if len(allowed) >= 2:
            assert not set(seen[0]) & set(seen[1])
        os.sched_setaffinity(0, before)
        assert bench.pin_rank(None, [], 0, 4, nodes)["cpus"] is None      # unknown node: left alone
        assert sorted(os.sched_getaffinity(0)) == allowed
    finally:
        os.sched_setaffinity(0, before)
