cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_zc; mkdir -p $O
for w in C2 C4; do for cfg in "wave auto 0" "wave 0 0" "lane auto 0" "lane 0 0" "lane 2 0" "wave 0 4" "wave auto 4"; do
  set -- $cfg
  key=${w}_$1_$2_$3
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$key -o t -- python $R/seq-align_amd/tools/walk_zc_ab.py $w $1 $2 $3 > $O/$key.log 2>&1
  f=$(find $O/$key -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$(grep -h median $O/$key.log | tail -1)" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
out=[]
for r in rows:
    n=r["Name"]
    if "traceback" in n or "fill_" in n or "Copy" in n or "copy" in n:
        out.append("%s %.1f us x%s"%(n.split("(")[0].replace("void sa::","")[:40], float(r["AverageNs"])/1e3, r["Calls"]))
print(sys.argv[2], "|", " | ".join(out))
PY
done; done
