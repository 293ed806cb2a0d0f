# Kernel / copy durations of seqalign_nw_batch (125 k pairs) phase by phase of seq-align_amd/tools/e2e_probe.py
# (fresh context, after torch.cuda init, after the placed arenas exist, ...): which of them change.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/e2eprobe -o t -- python $R/seq-align_amd/tools/e2e_probe.py > $R/gpurun_out/e2eprobe.log 2>&1
grep -v amdgpu.ids $R/gpurun_out/e2eprobe.log | tail -12
python - $R/gpurun_out/e2eprobe <<'PY'
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-36:]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r["Direction"].replace("MEMORY_COPY_", "")))
ev.sort()
for name in ("fill_nw_dirs_x2_kernel<3, 512>", "traceback_nw_dirs_kernel", "copy HOST_TO_DEVICE", "__amd_rocclr_copyBuffer"):
    d = [(e - s) / 1e3 for s, e, n in ev if n.endswith(name) and (e - s) > 50000]
    per = 64 if "fill" in name else (16 if "traceback_nw" in name else 64)
    print(name, "avg us per consecutive %d:" % per, " ".join("%.0f" % (sum(d[i:i + per]) / len(d[i:i + per])) for i in range(0, len(d), per)))
PY
