# kernel times of the BLOSUM62 fills by batch size (run on the GPU box from the repo root) -> stdout
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c4size -o t -- python $R/seq-align_amd/tools/c4_by_batch_size.py > $R/gpurun_out/c4size.log 2>&1
grep "pairs" $R/gpurun_out/c4size.log
python - $R/gpurun_out/c4size <<'PY'
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "fill_sw_best_x2" in n or "fill_dirs_x2" in n or "sw_sweep_dirs_ev" in n:
            rows.append((int(r["Start_Timestamp"]), n.split("(")[0].replace("void sa::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))))
rows.sort()
# VALU instructions per PAIR and cycles per instruction of each kernel (profiles/r05/r05c_e2e_pmc_insts.json, r05_valu_mix.json)
per_pair = {"fill_sw_best_x2_kernel<5, 1, 1024>": (182808000 / 4000, 3.368), "fill_dirs_x2_kernel<5, 1, 1024>": (191900000 / 4000, 3.493),
            "sw_sweep_dirs_ev_kernel<5, unsigned int>": (223636154 / 4000, 3.607)}
by = defaultdict(list)
for _, name, ms, grid in rows:
    by[(name, grid)].append(ms)
for (name, grid), v in sorted(by.items()):
    v = sorted(v)[: max(1, len(v) - 1)]            # (drop the first, slowest launch of each size)
    ms = sum(v) / len(v)
    waves = grid // 64 if "sweep" in name else grid // 64
    pairs = waves if "sweep" in name else waves * 2
    ip, cpi = per_pair.get(name, (0, 0))
    frac = ip * pairs * cpi / (1024 * 2.4e9 * ms * 1e-3) if ip else 0
    print(f"{name:45s} grid {grid:8d} (~{pairs} pairs): {ms:.4f} ms   frac of VALU issue {frac:.3f}   waves per SIMD {waves / 1024:.1f}")
PY
