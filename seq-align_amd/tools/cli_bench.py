#!/usr/bin/env python3
"""The command-line tool end to end on a big file: BASELINE configs[4]'s pair stream (C5's generator,
150 x 150 DNA, seed 5) written as FASTA, then `seqalign_nw --file` with its output to a file on the same disk -- wall clock,
pairs per second -- beside what the library call alone takes on the same pairs.

    python seq-align_amd/tools/cli_bench.py [pairs = 1 000 000]
"""
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "seq-align_amd" / "python"))
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from seqalign_amd import workloads as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tmp = Path("/tmp/cli_bench")
tmp.mkdir(exist_ok=True)
fa = tmp / f"c5_{n}.fa"
t0 = time.perf_counter()
with open(fa, "wb") as f:
    for lo in range(0, n, 50000):
        b = W.dna_nw_indexed(lo, min(50000, n - lo), seed=5)
        out = []
        for p in range(b.n_pairs):
            out.append(b">a%d\n%s\n>b%d\n%s\n" % (lo + p, b.seq_a(p), lo + p, b.seq_b(p)))
        f.write(b"".join(out))
print(f"wrote {fa} ({fa.stat().st_size / 1e6:.0f} MB, {n} pairs) in {time.perf_counter() - t0:.1f} s", flush=True)
import os
exe = Path(os.environ.get("CLI_EXE") or (ROOT / "seq-align_amd" / "bin" / "seqalign_nw"))
for rep in range(3):
    out = tmp / "out.txt"
    t0 = time.perf_counter()
    with open(out, "wb") as fo:
        subprocess.run([str(exe), "--printscores", "--file", str(fa)], stdout=fo, check=True, env=dict(os.environ, SEQALIGN_CLI_TIMING="1"))
    dt = time.perf_counter() - t0
    print(f"seqalign_nw --printscores --file: {dt:.3f} s wall, {n / dt / 1e6:.3f} M pairs/s, {n * 22500 / dt / 1e9:.1f} GCUPS end to end "
          f"(output {out.stat().st_size / 1e6:.0f} MB)", flush=True)
# the same file through `cat` (what reading it costs at all) and the library call alone on 125 k of the pairs
t0 = time.perf_counter()
subprocess.run(["cat", str(fa)], stdout=subprocess.DEVNULL, check=True)
print(f"cat of the input: {time.perf_counter() - t0:.3f} s", flush=True)
import seqalign_amd as S  # noqa: E402
ctx = S.Context(0)
sc = S.make_scoring({"preset": "default"})
b = W.dna_nw_indexed(0, 125000, seed=5)
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    ctx.nw_batch(b, sc, raw=True)
    ts.append(time.perf_counter() - t0)
print(f"seqalign_nw_batch on 125 000 of the pairs: {np.median(ts[1:]) * 1e3:.2f} ms -> {n / 125000 * np.median(ts[1:]):.3f} s for {n}", flush=True)

# ---- the same for seqalign_sw: BASELINE configs[2]'s reads (150 bp against 1 000 bp windows) as a file, the tool's defaults
# (--minscore from the lengths, every hit) and --maxhits 1
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
fs = tmp / f"c3_{ns}.fa"
with open(fs, "wb") as f:
    for lo in range(0, ns, 20000):
        b = W.dna_sw_read_vs_ref(min(20000, ns - lo), seed=2 + lo)
        f.write(b"".join(b">r%d\n%s\n>w%d\n%s\n" % (lo + p, b.seq_a(p), lo + p, b.seq_b(p)) for p in range(b.n_pairs)))
print(f"wrote {fs} ({fs.stat().st_size / 1e6:.0f} MB, {ns} read / window pairs)", flush=True)
exe_sw = exe.with_name("seqalign_sw")
for extra in ([], ["--maxhits", "1"]):
    for rep in range(2):
        out = tmp / "out_sw.txt"
        t0 = time.perf_counter()
        with open(out, "wb") as fo:
            subprocess.run([str(exe_sw), *extra, "--file", str(fs)], stdout=fo, check=True, env=dict(os.environ, SEQALIGN_CLI_TIMING="1"))
        dt = time.perf_counter() - t0
        print(f"seqalign_sw {' '.join(extra)} --file: {dt:.3f} s wall, {ns / dt / 1e6:.3f} M pairs/s, {ns * 151 * 1001 / dt / 1e9:.1f} GCUPS end to end "
              f"(output {out.stat().st_size / 1e6:.0f} MB)", flush=True)
