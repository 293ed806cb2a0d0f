# the local tile walker's tile edge: 64 x 64 bytes (32 lines per reload) against 32 x 32 (8 lines), kernel means and wall clock
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_tile; mkdir -p $O
for rep in 1 2; do for w in C2 C3 C4; do for t in 64 32; do
  key=${w}_tile${t}_r${rep}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$key -o t -- python $R/seq-align_amd/tools/ab_local.py $w 1 15 walk_tile=$t > $O/$key.log 2>&1
  f=$(find $O/$key -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$(grep -h median $O/$key.log | tail -1)" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))) if sys.argv[1] else []
out=[]
for r in rows:
    n=r["Name"]
    if ("traceback" in n) and int(r["Calls"]) > 5:
        out.append("%s %.1f us x%s"%(n.split("(")[0].replace("void sa::","")[:60], float(r["AverageNs"])/1e3, r["Calls"]))
print(sys.argv[2].split("launched")[0][:110], "|", " | ".join(out))
PY
  rm -rf $O/$key
done; done; done
