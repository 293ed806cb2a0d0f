# SQ wait / issue counters of the C4 fills for two builds of the library (same box): what a wave's cycles go to, before and after
# the LDS pipelining (round 6).   bash profiles/scripts/r06_pmc_fills.sh [libs...]    -> gpurun_out/r06_pmc_fills/summary.json
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_pmc_fills; mkdir -p $OUT
LIBS=${*:-libseqalign_hip.so libseqalign_hip_exp_off.so}
cd /tmp && export TMPDIR=/tmp
for lib in $LIBS; do for h in 1 4; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_IFETCH"; do
    i=$((i+1))
    SEQALIGN_LIB=$R/seq-align_amd/lib/$lib timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/$lib.h$h.pmc$i -o p -- python $R/seq-align_amd/tools/sw_enum_profile.py C4 $h > $OUT/$lib.h$h.pmc$i.log 2>&1
  done
done; done
python - $OUT <<'PY' | tee $OUT/summary.txt
import glob, os, sqlite3, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for db in glob.glob(os.path.join(out, "**", "*.db"), recursive=True):
    lib = os.path.relpath(db, out).split(".h")[0]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    for name, ctr, val in c.execute(f"select {kn}, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name"):
        if "fill_" in name:
            acc[(lib, name.split("(")[0].replace("void sa::", ""))][ctr].append(float(val))
for (lib, k), cs in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k:38s} {lib:28s} wave_cycles {wc/1e6:7.1f}M  active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:.3f}  wait_any {m.get('SQ_WAIT_ANY',0)/wc:.3f}  wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:.3f}"
          f"  wait_lds {m.get('SQ_WAIT_INST_LDS',0)/wc:.3f}  valu_act {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.3f} lds_act {m.get('SQ_ACTIVE_INST_LDS',0)/wc:.3f} sca_act {m.get('SQ_ACTIVE_INST_SCA',0)/wc:.3f}"
          f"  insts valu {m.get('SQ_INSTS_VALU',0)/1e6:.1f}M salu {m.get('SQ_INSTS_SALU',0)/1e6:.1f}M lds {m.get('SQ_INSTS_LDS',0)/1e6:.1f}M br {m.get('SQ_INSTS_BRANCH',0)/1e6:.1f}M  bank_conf {m.get('SQ_LDS_BANK_CONFLICT',0)/1e6:.1f}M")
PY
