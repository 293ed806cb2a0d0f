#!/usr/bin/env python3
"""HERE: put the headline kernel's PMC traffic of `profiles/r05/<tag>_C2.json` (profiles/collect.sh) into profiles/pmc_traffic.json,
the static table bench.py quotes as roofline.traffic (PMC counters cannot be read from inside the benchmarked process)."""
import csv, json, re, sys
tag = sys.argv[1]
src = f"profiles/r05/{tag}_C2.json"
d = json.load(open(src))
k = next(n for n in d["pmc_per_launch"] if n.startswith("fill_stream_kernel"))
p = d["pmc_per_launch"][k]
# the kernel's average duration: the --kernel-trace --stats pass (the PMC passes slow the kernels down)
stats_avg_us = next(float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f"profiles/r05/{tag}_C2_kernel_stats.csv")) if "fill_stream_kernel" in r["Name"])
cmd = re.sub(r"\S*/bench\.py", "bench.py", d["command"])
t = json.load(open("profiles/pmc_traffic.json"))
t["C2:stream:10000"] = {
    "hbm_bytes_per_launch": p["hbm_bytes_per_launch"], "write_bytes": p["write_bytes"], "fetch_bytes_corrected": p["fetch_bytes_corrected"],
    "kernel_avg_us": stats_avg_us,
    "source": f"{src} (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate passes of `{cmd}`, round 5, the final "
              "tree; FETCH_SIZE x2 per MI355X_MICROARCH.md; WRITE_SIZE calibrated 1.00x on torch fill_, profiles/r01_rowscan_c3.json)"}
json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(t["C2:stream:10000"], indent=1))
