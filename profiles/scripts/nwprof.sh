cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /root/repo/gpurun_out/nwprof -o s -- python /root/repo/seq-align_amd/tools/nw_profile.py ${NW_PAIRS:-10000} > /root/repo/gpurun_out/nwprof.log 2>&1
grep nw_batch /root/repo/gpurun_out/nwprof.log
python - <<'PY'
import csv,glob
for pat in ("*kernel_stats.csv","*memory_copy_stats.csv"):
    for f in glob.glob("/root/repo/gpurun_out/nwprof/**/"+pat, recursive=True):
        for r in list(csv.DictReader(open(f)))[:8]:
            print(r["Name"][:60], r["Calls"], round(float(r["TotalDurationNs"])/1e6,3), "ms total", round(float(r["AverageNs"])/1e3,1), "us avg")
PY
