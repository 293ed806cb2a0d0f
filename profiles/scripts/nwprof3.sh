cd /tmp && export TMPDIR=/tmp
for n in 10000 125000; do
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /root/repo/gpurun_out/nwprof3_$n -o s -- python /root/repo/seq-align_amd/tools/nw_profile.py $n > /root/repo/gpurun_out/nwprof3_$n.log 2>&1
  grep nw_batch /root/repo/gpurun_out/nwprof3_$n.log
  python - /root/repo/gpurun_out/nwprof3_$n <<'PY'
import csv,glob,sys
for pat in ("*kernel_stats.csv","*memory_copy_stats.csv"):
    for f in glob.glob(sys.argv[1]+"/**/"+pat, recursive=True):
        for r in list(csv.DictReader(open(f)))[:8]:
            print(r["Name"][:70], r["Calls"], round(float(r["TotalDurationNs"])/1e6,3), "ms total", round(float(r["AverageNs"])/1e3,1), "us avg")
PY
done
