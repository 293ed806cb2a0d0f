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
python "$R/profiles/e2e_roofline_summarise.py" "$OUT" "$R" "$TAG" > "$OUT/e2e_roofline.json"
cat "$OUT/e2e_roofline.json" | head -60
