# PMC passes over seqalign_sw_batch (up to 4 hits) for the sweep and traceback kernels: instruction mix, issue / wait split.
# Run ON THE GPU BOX from the repo root: bash profiles/scripts/swpmc.sh  -> gpurun_out/swpmc_<cfg>.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in ${SW_CFGS:-C3 C4}; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
             "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
             "FETCH_SIZE" \
             "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/swpmc_$w/p$i -o p -- python $R/seq-align_amd/tools/sw_enum_profile.py $w 4 > $R/gpurun_out/swpmc_$w.p$i.log 2>&1
  done
  python - $R/gpurun_out/swpmc_$w $w <<'PY' > $R/gpurun_out/swpmc_$w.json
import glob, json, sqlite3, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    for name, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
        for k in ("sw_sweep_kernel", "sw_sweep_dirs_kernel", "fill_dirs_kernel", "fill_dirs_x2_kernel", "fill_sw_best_x2_kernel", "traceback_dirs_kernel", "traceback_dirs_tile_kernel", "fill_stream_kernel"):
            if k in name:
                acc[name.split("(")[0][-70:]][cname].append(float(val))
out = {"workload": sys.argv[2], "command": "seq-align_amd/tools/sw_enum_profile.py %s 4 (3 calls)" % sys.argv[2], "per_launch": {}}
for k, cs in acc.items():
    out["per_launch"][k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out["per_launch"][k]["launches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))
PY
  cat $R/gpurun_out/swpmc_$w.json
done
