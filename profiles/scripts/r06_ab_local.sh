# the direction byte's local form against the older one: kernel means (rocprofv3 --kernel-trace --stats) and the calls' wall clock
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_local; mkdir -p $O
for rep in 1 2; do for w in C2 C3 C4 C5; do for l in 0 1; do
  key=${w}_local${l}_r${rep}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$key -o t -- python $R/seq-align_amd/tools/ab_local.py $w $l > $O/$key.log 2>&1
  f=$(find $O/$key -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$(grep -h median $O/$key.log | tail -1)" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))) if sys.argv[1] else []
out=[]
for r in rows:
    n=r["Name"]
    if "traceback" in n or "fill_" in n:
        out.append("%s %.1f us x%s"%(n.split("(")[0].replace("void sa::","")[:60], float(r["AverageNs"])/1e3, r["Calls"]))
print(sys.argv[2], "|", " | ".join(out))
PY
  rm -rf $O/$key
done; done; done
