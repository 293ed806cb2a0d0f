#!/usr/bin/env python3
"""HERE: put the fill kernel's PMC traffic of `profiles/<round>/<tag>_<W>.json` (profiles/collect.sh) into profiles/pmc_traffic.json,
the static table bench.py quotes as roofline.traffic (PMC counters cannot be read from inside the benchmarked process).
    python profiles/scripts/update_pmc_traffic.py <tag> [round dir = r06] [workloads = C2 C3 C4 C5]"""
import csv, json, re, sys
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r06"
workloads = sys.argv[3:] or ["C2", "C3", "C4", "C5"]
PAIRS = {"C2": 10000, "C3": 10000, "C4": 4000, "C5": 125000}
t = json.load(open("profiles/pmc_traffic.json"))
for w in workloads:
    src = f"profiles/{rnd}/{tag}_{w}.json"
    try:
        d = json.load(open(src))
    except OSError as ex:
        print(w, "skipped:", ex)
        continue
    k = next(n for n in d["pmc_per_launch"] if n.startswith("fill_stream_kernel"))
    p = d["pmc_per_launch"][k]
    # the kernel's average duration: the --kernel-trace --stats pass (the PMC passes slow the kernels down)
    stats_avg_us = next(float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f"profiles/{rnd}/{tag}_{w}_kernel_stats.csv"))
                        if "fill_stream_kernel" in r["Name"])
    cmd = re.sub(r"\S*/bench\.py", "bench.py", d["command"])
    key = f"{w}:stream:{PAIRS[w]}"
    t[key] = {
        "hbm_bytes_per_launch": p["hbm_bytes_per_launch"], "write_bytes": p["write_bytes"], "fetch_bytes_corrected": p["fetch_bytes_corrected"],
        "kernel_avg_us": stats_avg_us,
        "source": f"{src} (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate passes of `{cmd}`, round {rnd[1:].lstrip('0')}, this "
                  "tree; FETCH_SIZE x2 per MI355X_MICROARCH.md; WRITE_SIZE calibrated 1.00x on torch fill_, profiles/r01_rowscan_c3.json)"}
    print(key, json.dumps(t[key], indent=1))
json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)
