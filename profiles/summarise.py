#!/usr/bin/env python3
"""Summarise one profiles/collect.sh output directory into JSON on stdout.

Reads the rocprofv3 sqlite outputs (*_results.db: views `kernels`,
`counters_collection`, `top_kernels`).  Per kernel of ours: launches and mean
duration from the kernel trace, and each PMC counter averaged per launch.

HBM bytes follow MI355X_MICROARCH.md's HBM section: WRITE_SIZE / FETCH_SIZE are
in KiB and derive from the L2's memory-side request counters; on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x (doubled here as
`fetch_bytes_corrected`); WRITE_SIZE is uncalibrated in that guide, so
collect.sh CALIB=1 calibrates it on torch fill_/copy_ kernels of known size."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

out_dir = sys.argv[1]
res = {"dir": os.path.basename(out_dir)}
try:
    res["command"] = open(os.path.join(out_dir, "command.txt")).read().strip()
except OSError:
    pass

OURS = ("fill_rowscan", "fill_wavefront", "fill_stream", "fill_wgstream", "fill_strips", "sw_reduce", "sw_enumerate", "traceback")


def short(name):
    for k in OURS:
        if k in name:
            return name[name.find(k):].split("(")[0][:80]
    return None


def dbs(sub):
    return glob.glob(os.path.join(out_dir, sub, "**", "*.db"), recursive=True)


dur = defaultdict(list)
top = []
for f in dbs("trace"):
    con = sqlite3.connect(f)
    for name, d in con.execute("select name, duration from kernels"):
        s = short(name)
        if s:
            dur[s].append(d / 1e3)
    try:
        top = [dict(name=n[:70], calls=c, total_us=t / 1e3, avg_us=a / 1e3, pct=p)
               for n, c, t, a, p in con.execute(
                   "select name,total_calls,total_duration,average,percentage from top_kernels limit 8")]
    except sqlite3.Error:
        pass
res["kernel_trace_us"] = {k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
                          for k, v in dur.items()}
res["top_kernels"] = top

pmc = defaultdict(lambda: defaultdict(list))
for f in dbs("pmc*"):
    con = sqlite3.connect(f)
    for name, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
        s = short(name)
        if s:
            pmc[s][cname].append(float(val))
res["pmc_per_launch"] = {}
for k, cs in pmc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    if "WRITE_SIZE" in d:
        d["write_bytes"] = d["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in d:
        d["fetch_bytes_raw"] = d["FETCH_SIZE"] * 1024
        d["fetch_bytes_corrected"] = d["FETCH_SIZE"] * 1024 * 2
    if "write_bytes" in d and "fetch_bytes_corrected" in d:
        d["hbm_bytes_per_launch"] = d["write_bytes"] + d["fetch_bytes_corrected"]
    res["pmc_per_launch"][k] = d

cal = {}
for tag, ctr in (("calib_w", "WRITE_SIZE"), ("calib_r", "FETCH_SIZE")):
    acc = defaultdict(list)
    for f in dbs(tag):
        con = sqlite3.connect(f)
        for name, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
            if cname == ctr:
                acc[name[:70]].append(float(val) * 1024)
    if acc:
        cal[ctr] = {k: {"launches": len(v), "median_bytes": sorted(v)[len(v) // 2]} for k, v in acc.items()}
if cal:
    cal["buffer_bytes"] = 2_739_120_000 // 4 * 4
    res["calibration"] = cal
print(json.dumps(res, indent=1))
