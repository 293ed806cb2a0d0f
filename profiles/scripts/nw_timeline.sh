#!/bin/bash
# Kernel timeline of the LAST seqalign_nw_batch call of tools/nw_profile.py <pairs>, per option variant (run on the GPU box):
#   bash profiles/scripts/nw_timeline.sh 125000 "SEQALIGN_WALK_OVERLAP=1" "SEQALIGN_WALK_OVERLAP=0"
R=${GRAFT_REPO_ROOT:-$PWD}
N=$1; shift
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  D=/tmp/nwtl_$$_$(echo "$V" | tr -c 'A-Za-z0-9' '_')
  env $V timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/seq-align_amd/tools/nw_profile.py $N > $D.log 2>&1
  echo "== $V  (host: $(grep nw_batch $D.log | tail -3 | awk '{print $4}' | tr '\n' ' ') ms)"
  python3 - "$D" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sa::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sa::", "")[:44], r.get("Queue_Id", "")))
rows.sort()
# the last call: kernels after the last gap of > 2 ms
cut = 0
for k in range(1, len(rows)):
    if rows[k][0] - rows[k - 1][1] > 1_000_000: cut = k
t0 = rows[cut][0]
for s, e, n, q in rows[cut:]:
    print(f"   {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{q}  {n}")
PY
done
