# Timeline of the last seqalign_nw_batch call of seq-align_amd/tools/e2e_probe.py steps <variant> (kernels and copies in time order)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
v=${1:-only_stream}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/probe3_$v -o t -- python $R/seq-align_amd/tools/e2e_probe.py steps $v > $R/gpurun_out/probe3_$v.log 2>&1
grep -E "^$v" $R/gpurun_out/probe3_$v.log
python - $R/gpurun_out/probe3_$v <<'PY'
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r["Direction"].replace("MEMORY_COPY_", "")))
ev.sort()
i0 = len(ev) - 1
while i0 > 0 and ev[i0][0] - max(e[1] for e in ev[:i0]) < 600000: i0 -= 1
t0 = ev[i0][0]
for s, e, n in ev[i0:]:
    print("%9.3f %9.3f  %8.1f us  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, n))
PY
