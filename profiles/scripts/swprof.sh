cd /tmp && export TMPDIR=/tmp
for w in C3 C4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/swenum_$w -o s -- python /root/repo/seq-align_amd/tools/sw_enum_profile.py $w 4 > /root/repo/gpurun_out/swenum_$w.log 2>&1
  grep -v "^W2026\|amdgpu.ids" /root/repo/gpurun_out/swenum_$w.log | tail -5
  python - /root/repo/gpurun_out/swenum_$w <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:70], r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), "ms total", round(float(r["AverageNs"])/1e3,1), "us avg")
PY
done
