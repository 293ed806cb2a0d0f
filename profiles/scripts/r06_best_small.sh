# the SW best-hit fill between one and two rounds of four-per-wave waves (4 096 .. 8 191 pairs, 150 x 1 000): the shipped rule (two per wave)
# against an experiment build that goes four per wave / mixed from 4 096 pairs (make exp EXPFLAGS=-DSA_BEST_X4_MIN=4096u)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_bestsmall; mkdir -p $O
for n in 4096 5000 6144 7000 8000; do for lib in main exp; do
  key=n${n}_${lib}
  L=$R/seq-align_amd/lib/libseqalign_hip.so; [ $lib = exp ] && L=$R/seq-align_amd/lib/libseqalign_hip_exp.so
  SEQALIGN_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$key -o t -- python $R/seq-align_amd/tools/ab_local.py C3 1 8 pairs=$n > $O/$key.log 2>&1
  f=$(find $O/$key -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$n $lib $(grep -h median $O/$key.log | tail -1 | sed 's/.*median/median/' | cut -c1-40)" <<'PY'
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
