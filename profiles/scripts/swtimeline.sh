# Timeline of the last seqalign_sw_batch call (kernels and copies in time order).  Run ON THE GPU BOX from the repo root:
#   SW_CFG=C3 SW_HITS=4 bash profiles/scripts/swtimeline.sh   -> gpurun_out/swtimeline_<cfg>_<hits>/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
w=${SW_CFG:-C3}; h=${SW_HITS:-4}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/swtimeline_${w}_$h -o t -- python $R/seq-align_amd/tools/sw_enum_profile.py $w $h > $R/gpurun_out/swtimeline_${w}_$h.log 2>&1
grep max_hits $R/gpurun_out/swtimeline_${w}_$h.log
python - $R/gpurun_out/swtimeline_${w}_$h <<'PY'
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:] + " q" + r["Queue_Id"]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r["Direction"].replace("MEMORY_COPY_", "")))
ev.sort()
i0 = len(ev) - 1
while i0 > 0 and ev[i0][0] - max(e[1] for e in ev[:i0]) < 1000000: i0 -= 1
t0 = ev[i0][0]
for s, e, n in ev[i0:]:
    print("%9.3f %9.3f  %8.1f us  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, n))
PY
