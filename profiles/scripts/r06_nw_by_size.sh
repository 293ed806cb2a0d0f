# the NW direction fill by batch size (BASELINE configs[1]'s shape, 150 x 150): two per wave (quad=1), the library's choice (quad=0), four per wave (quad=2)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_nwsize; mkdir -p $O
for n in 2048 3000 4096 5000 6144 7000 8192 10000 12288 14000 16384; do for q in 1 0 2; do
  key=n${n}_q${q}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$key -o t -- python $R/seq-align_amd/tools/ab_local.py C2 1 8 quad=$q pairs=$n > $O/$key.log 2>&1
  f=$(find $O/$key -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$n $q $(grep -h median $O/$key.log | tail -1 | sed 's/.*median/median/' | cut -c1-40)" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))) if sys.argv[1] else []
out=[]
for r in rows:
    n=r["Name"]
    if "fill_" in n and int(r["Calls"]) > 5:
        out.append("%s %.1f us"%(n.split("(")[0].replace("void sa::","")[:50], float(r["AverageNs"])/1e3))
print(sys.argv[2], "|", " | ".join(out))
PY
  rm -rf $O/$key
done; done
