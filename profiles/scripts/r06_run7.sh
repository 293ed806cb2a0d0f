cd $GRAFT_REPO_ROOT
echo "== legacy single-pair latency, stage laps (SEQALIGN_TIMING=1), last calls of each size"
SEQALIGN_TIMING=1 python seq-align_amd/tools/legacy_latency.py 2> gpurun_out/r06_legacy_laps.txt | tail -12
python - <<'PY'
import re, statistics
from collections import defaultdict
laps = defaultdict(list); order = []
blocks = []; cur = []
for line in open("gpurun_out/r06_legacy_laps.txt"):
    m = re.match(r"\[seqalign timing\] (.*?)\s+([0-9.]+) ms", line)
    if not m: continue
    k, v = m.group(1).strip(), float(m.group(2)) * 1e3
    cur.append((k, v))
    if k.startswith("one pair: matrices out"):
        blocks.append(cur); cur = []
n = len(blocks) // 3
for name, part in (("9x10", blocks[4:n]), ("150x150", blocks[n + 4:2 * n]), ("150x1000", blocks[2 * n + 4:])):
    acc = defaultdict(list)
    for b in part:
        for k, v in b: acc[k].append(v)
    print(name, {k: round(statistics.median(v), 1) for k, v in acc.items()}, "us (medians)")
PY
echo "== unplaced arenas: the streams de-phased in the kernel (SA_EXP_DEPHASE) against the product, alternating"
for rep in 1 2 3; do for lib in libseqalign_hip.so libseqalign_hip_exp_dephase.so; do
  SEQALIGN_LIB=$PWD/seq-align_amd/lib/$lib python bench.py --no-cpu-baseline --no-e2e --no-configs --steps 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); rf=r['roofline']; print('$lib', 'placed frac %.3f kernel_ms %.4f  unplaced frac %.3f kernel_ms %.4f  bit_exact %s' % (rf['frac'], rf['kernel_ms'], rf.get('frac_unplaced',0), rf['unplaced']['kernel_ms'], r['bit_exact_vs_oracle']))"
done; done
echo "== FETCH_SIZE / WRITE_SIZE of the walkers (C2: seqalign_nw_batch, 10 000 walks)"
cd /tmp; export TMPDIR=/tmp
for set in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/r06_walk_pmc_${set%% *} -o p -- python $GRAFT_REPO_ROOT/seq-align_amd/tools/walk_group_ab.py C2 > /dev/null 2>&1
done
python $GRAFT_REPO_ROOT/profiles/scripts/pmc_db_summary.py $GRAFT_REPO_ROOT/gpurun_out traceback 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    print(k[:60], {c: round(x['mean'],1) for c,x in v.items()})"
