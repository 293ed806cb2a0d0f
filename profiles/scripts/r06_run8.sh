cd $GRAFT_REPO_ROOT
echo "== full GPU suite on the blocked direction bytes"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6
echo "== x2_check 45 s, fuzz_e2e 60 s"
timeout 400 python seq-align_amd/tools/x2_check.py 45 2>&1 | grep -v amdgpu.ids | tail -4
timeout 400 python seq-align_amd/tools/fuzz_e2e.py --seconds 60 --seed 6101 2>&1 | grep -v amdgpu.ids | tail -2
echo "== A/B blocked (product) vs row-major (exp): kernel stats"
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06_blk; mkdir -p $O
for rep in 1 2; do for lib in libseqalign_hip.so libseqalign_hip_exp_rowmajor.so; do
  for w in "sw C4 1" "sw C3 1" "nw 10000" "nw 125000"; do
    set -- $w
    if [ $1 = sw ]; then cmd="python $GRAFT_REPO_ROOT/seq-align_amd/tools/sw_enum_profile.py $2 $3"; key=$2_$3; else cmd="python $GRAFT_REPO_ROOT/seq-align_amd/tools/nw_profile.py $2"; key=nw_$2; fi
    SEQALIGN_LIB=$GRAFT_REPO_ROOT/seq-align_amd/lib/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$lib.$key.$rep -o t -- $cmd > $O/$lib.$key.$rep.log 2>&1
    f=$(find $O/$lib.$key.$rep -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$rep $lib $key: $(grep -h ' ms' $O/$lib.$key.$rep.log | tail -1)" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
out=[]
for r in rows:
    n=r["Name"]
    if "fill_" in n or "traceback" in n:
        out.append("%s %.1f us x%s"%(n.split("(")[0].replace("void sa::","")[:44], float(r["AverageNs"])/1e3, r["Calls"]))
print(sys.argv[2], "|", " | ".join(out))
PY
  done
done; done
