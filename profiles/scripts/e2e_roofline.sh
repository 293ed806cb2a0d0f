#!/bin/bash
# The `valu_issue` roofline of the host-level calls' dominant kernels (bench.py: e2e.roofline).  Run ON THE GPU BOX from the
# repo root:   bash profiles/scripts/e2e_roofline.sh <tag>      -> gpurun_out/e2e_roofline_<tag>/{e2e_roofline.json, *_kernel_stats.csv}
# Per workload: one rocprofv3 --kernel-trace --stats pass (durations) and one --pmc SQ_INSTS_VALU pass (its own run: counters
# never share a run with other trace domains) over the host-level call, then
#     frac = SQ_INSTS_VALU x (cycles per instruction of the kernel's hot loop, static: profiles/r04/r04_valu_mix.json, from the
#            rate classes of profiles/r03/r03_valu_rate_probe.txt) / (1024 SIMDs x 2.4 GHz x kernel duration)
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/e2e_roofline_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # key, command...
  key=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$key.trace" -o t -- "$@" > "$OUT/$key.trace.log" 2>&1
  find "$OUT/$key.trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${key}_kernel_stats.csv" \;
  find "$OUT/$key.trace" -name "*kernel_trace.csv" -exec cp {} "$OUT/${key}_kernel_trace.csv" \;
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d "$OUT/$key.pmc" -o p -- "$@" > "$OUT/$key.pmc.log" 2>&1
}
run C2_10000 python $R/seq-align_amd/tools/nw_profile.py 10000
run C5_125000 python $R/seq-align_amd/tools/nw_profile.py 125000
run C3_10000 python $R/seq-align_amd/tools/sw_enum_profile.py C3 1
run C4_4000 python $R/seq-align_amd/tools/sw_enum_profile.py C4 1
run C3_10000_hits4 python $R/seq-align_amd/tools/sw_enum_profile.py C3 4
run C4_4000_hits4 python $R/seq-align_amd/tools/sw_enum_profile.py C4 4
python - "$OUT" "$R" "$TAG" <<'PY' > "$OUT/e2e_roofline.json"
import csv, glob, json, os, sqlite3, sys
from collections import defaultdict
out, root, tag = sys.argv[1:4]
mix = json.load(open(os.path.join(root, "profiles", "r04", "r04_valu_mix.json")))
N_SIMD, CLOCK = 1024, 2.4e9
res = {}
for key in ("C2_10000", "C5_125000", "C3_10000", "C4_4000", "C3_10000_hits4", "C4_4000_hits4"):
    dur = defaultdict(list)
    f = os.path.join(out, key + "_kernel_trace.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "sa::" in n:
            dur[n.split("(")[0].replace("void ", "").strip()].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    if not dur:
        continue
    insts = defaultdict(list)
    for db in glob.glob(os.path.join(out, key + ".pmc", "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        for name, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
            if cname == "SQ_INSTS_VALU" and "sa::" in name:
                insts[name.split("(")[0].replace("void ", "").strip()].append(float(val))
    kernels = {}
    for k, v in dur.items():
        v = v[len(v) // 3:]          # the first call sizes buffers: later launches only
        e = {"launches_seen": len(v), "kernel_ms": sum(v) / len(v), "total_ms_share": None}
        iv = insts.get(k)
        if iv:
            e["instructions"] = sum(iv) / len(iv)
        m = next((x for x in mix if x["demangled"] in k.replace("sa::", "")), None)
        if m and iv:
            cpi = m["mix"]["cycles_per_valu_instruction"]
            e["cycles_per_instruction"] = cpi
            e["frac"] = e["instructions"] * cpi / (N_SIMD * CLOCK * e["kernel_ms"] * 1e-3)
        kernels[k] = e
    total = sum(e["kernel_ms"] * e["launches_seen"] for e in kernels.values())
    for e in kernels.values():
        e["total_ms_share"] = round(e["kernel_ms"] * e["launches_seen"] / total, 3)
    dom = max(kernels, key=lambda k: kernels[k]["kernel_ms"] * kernels[k]["launches_seen"])
    d = kernels[dom]
    res[key] = {"bound": "valu_issue", "kernel": dom, "kernel_ms": round(d["kernel_ms"], 4),
                "instructions": d.get("instructions"), "cycles_per_instruction": d.get("cycles_per_instruction"),
                "frac": round(d["frac"], 3) if "frac" in d else None,
                "peak": "1024 SIMDs x 2.4 GHz; cycles per wave64 instruction by rate class (profiles/r03/r03_valu_rate_probe.txt) "
                        "weighted by the static mix of the kernel's row loop (profiles/r04/r04_valu_mix.json)",
                "source": f"profiles/{tag}/: rocprofv3 --kernel-trace (durations) and --pmc SQ_INSTS_VALU (own pass) over the host-level call",
                "kernels_of_the_call": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in e.items()} for k, e in kernels.items()}}
print(json.dumps(res, indent=1))
PY
cat "$OUT/e2e_roofline.json" | head -60
