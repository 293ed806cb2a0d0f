#!/bin/bash
# What the waves of the multi-hit path's kernels wait for (sw_sweep_dirs_ev_kernel at 0.42 of VALU issue).
# Run ON THE GPU BOX from the repo root:  bash profiles/scripts/sweep_pmc.sh <tag> [C3|C4]  -> gpurun_out/sweep_pmc_<tag>/
# One rocprofv3 --kernel-trace --stats pass, then SQ counter groups, each in its own run (counters never share a run with other
# trace domains); summarised per kernel (mean over the launches seen).
TAG=${1:-r05}
WL=${2:-C3}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/sweep_pmc_${TAG}_$WL
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/seq-align_amd/tools/sw_enum_profile.py $WL 4"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.log" 2>&1
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc$i" -o p -- $CMD > "$OUT/pmc$i.log" 2>&1
done
python - "$OUT" <<'PY' > "$OUT/summary.json"
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, cs in acc.items():
    if not any(x in k for x in ("sweep", "fill_", "traceback", "sw_")):
        continue
    res[k] = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in sorted(cs.items())}
print(json.dumps(res, indent=1))
PY
head -c 6000 "$OUT/summary.json"
