#!/bin/bash
# profiles/collect.sh -- run ON THE GPU BOX (via gpurun) from the repo root.
#   bash profiles/collect.sh <tag> [bench args...]
# Collects, for `python bench.py <bench args>`:
#   1. rocprofv3 --kernel-trace --stats            (per-kernel durations)
#   2. rocprofv3 --pmc ... in SEPARATE passes      (HBM bytes, L2/TA/SQ state)
# into gpurun_out/prof_<tag>/, then summarises into gpurun_out/prof_<tag>/summary.json
# (copy the summaries you want judged into profiles/).  PMC passes never combine
# with trace domains other than the kernel trace.
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-unplaced --no-configs $*"
echo "$CMD" > "$OUT/command.txt"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.log" 2>&1
# the same pass once more as CSV: the human-readable --stats table that gets committed
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $CMD > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# PMC_SETS=traffic: only the two HBM-traffic passes (WRITE_SIZE, FETCH_SIZE; what bench.py's roofline.traffic quotes)
if [ "${PMC_SETS:-all}" = traffic ]; then
  i=0
  for set in "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc$i" -o p -- $CMD > "$OUT/pmc$i.log" 2>&1
  done
  python "$REPO/profiles/summarise.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/summarise.err"
  cat "$OUT/summary.json"
  exit 0
fi
i=0
for set in "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "FETCH_SIZE" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_WRITE_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_DRAM_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TOTAL_WRITE_sum TA_FLAT_WRITE_WAVEFRONTS_sum TD_STORE_WAVEFRONT_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc$i" -o p -- $CMD > "$OUT/pmc$i.log" 2>&1
done
python "$REPO/profiles/summarise.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/summarise.err"
cat "$OUT/summary.json"
# optional calibration of WRITE_SIZE/FETCH_SIZE on kernels of KNOWN traffic
# (torch fill_: 4n bytes written; copy_: 4n read + 4n written), CALIB=1
if [ "${CALIB:-0}" = 1 ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/calib_w" -o c -- python "$REPO/seq-align_amd/tools/hbm_ceiling.py" > "$OUT/calib_w.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/calib_r" -o c -- python "$REPO/seq-align_amd/tools/hbm_ceiling.py" > "$OUT/calib_r.log" 2>&1
  python - "$OUT" <<'PY' > "$OUT/calibration.json"
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
n_bytes = 2_739_120_000 // 4 * 4
res = {"buffer_bytes": n_bytes}
for tag, ctr in (("calib_w", "WRITE_SIZE"), ("calib_r", "FETCH_SIZE")):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == ctr:
                acc[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]) * 1024)
    res[ctr] = {k: {"launches": len(v), "median_bytes": sorted(v)[len(v) // 2]} for k, v in acc.items()}
print(json.dumps(res, indent=1))
PY
  cat "$OUT/calibration.json"
fi
