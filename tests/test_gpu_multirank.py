"""GPU tier: the multi-rank bench path on real hardware (BASELINE configs[4]).

The driver's scaling run is `bench.py --gpus N` on an 8-GPU node; on the 1-GPU box these tests run the SAME code with the
ranks folded onto device 0 (`config.ranks_share_devices: true`): rank launch, the key-value-store control plane, per-rank
arena placement, NUMA pinning, C5's index sharding, the MAX / SUM reductions and the one-line contract -- so that the
first real 8-GPU run is a repeat, not a premiere.  Both launch forms: bench.py's own launcher, and the driver's
`python -m torch.distributed.run` line.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
ARGS = ["--gpus", "2", "--pairs", "4000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR",
                                                            "TORCHELASTIC_USE_AGENT_STORE", "SEQALIGN_DIST_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # two ranks on one device: a short placement walk each (the walk is a per-process search over free HBM; folded ranks
    # would otherwise walk the same memory one after the other for most of the test's time)
    env.setdefault("SEQALIGN_ARENA_SCAN_GIB", "8")
    env.update(extra)
    return env


def _check(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[:800]          # ONE line on stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1
    assert res["metric"] == "dp_cell_updates_per_sec" and res["unit"] == "GCUPS" and res["scaling"] == "weak"
    assert res["bit_exact_vs_oracle"] is True
    cfg = res["config"]
    assert cfg["ranks_share_devices"] is True                    # 1-GPU box: the two ranks folded onto device 0
    assert cfg["workload"].startswith("C5") and cfg["pairs_per_gpu"] == 4000 and cfg["global_pairs"] == 8000
    assert "scale_base" in cfg and cfg["scale_base"]["per_gpu_alone_gcups"] > 1.0
    ranks = res["per_rank"]
    assert len(ranks) == 2 and sorted(r["rank"] for r in ranks) == [0, 1]
    assert all(r["kernel_ms"] > 0 for r in ranks)
    # whole-job value = both ranks' cells / max-rank time
    cells = 2 * 4000 * 150 * 150
    assert abs(res["value"] - cells * 3 / (res["ms_per_step"] * 3e-3) / 1e9) < 1e-6 * res["value"]
    assert res["value"] > 1.0                                    # north_star's floor, GCUPS
    assert res["roofline"]["bound"] == "hbm" and 0 < res["roofline"]["frac"] < 1
    return res


def test_bench_two_ranks_own_launcher():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), *ARGS], env=_clean_env(), capture_output=True, text=True,
                         timeout=900)
    _check(out)


def test_bench_two_ranks_under_torch_distributed_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), *ARGS]
    out = subprocess.run(cmd, env=_clean_env(OMP_NUM_THREADS="4"), capture_output=True, text=True, timeout=900)
    _check(out)
