# PMC passes over seqalign_nw_batch (direction-byte path) on C2: instruction mix, issue / wait split, HBM bytes of the fill and the walk.
# Run ON THE GPU BOX from the repo root: bash profiles/scripts/nwpmc.sh  -> gpurun_out/nwpmc.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/nwpmc/p$i -o p -- python $R/seq-align_amd/tools/nw_profile.py ${NW_PAIRS:-10000} > $R/gpurun_out/nwpmc.p$i.log 2>&1
done
python - $R/gpurun_out/nwpmc <<'PY' > $R/gpurun_out/nwpmc.json
import glob, json, sqlite3, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    for name, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
        for k in ("fill_nw_dirs_kernel", "fill_nw_dirs_x2_kernel", "traceback_nw_dirs_kernel", "traceback_dirs_tile_kernel"):
            if k in name:
                acc[name.split("(")[0][-70:]][cname].append(float(val))
out = {"workload": "C2 through seqalign_nw_batch", "command": "seq-align_amd/tools/nw_profile.py (6 calls)", "per_launch": {}}
for k, cs in acc.items():
    out["per_launch"][k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out["per_launch"][k]["launches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))
PY
cat $R/gpurun_out/nwpmc.json
