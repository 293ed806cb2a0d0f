# HERE (not on the GPU box): copy what `r05_records.sh <tag>` left under gpurun_out/ into profiles/r05/ (the tracked records)
TAG=${1:?tag}; D=profiles/r05
cp gpurun_out/$TAG/${TAG}_bench_*.json gpurun_out/$TAG/${TAG}_gputests.log gpurun_out/$TAG/${TAG}_C2.json gpurun_out/$TAG/${TAG}_C2_kernel_stats.csv $D/
for f in gpurun_out/e2e_roofline_$TAG/*_kernel_stats.csv; do cp $f $D/${TAG}_e2e_$(basename $f); done
cp gpurun_out/e2e_roofline_$TAG/e2e_pmc_insts.json $D/${TAG}_e2e_pmc_insts.json
cp gpurun_out/e2e_roofline_$TAG/e2e_roofline.json $D/${TAG}_e2e_roofline.json
cp gpurun_out/e2e_roofline_$TAG/e2e_roofline.json profiles/e2e_roofline.json
ls $D | grep "^$TAG" | wc -l
# the bench line the rocprof'd process itself printed (its live HIP-event figure belongs beside that process's kernel trace: every
# process finds its own placement, 0.82-0.86)
grep -h '"metric"' gpurun_out/prof_${TAG}_C2/stats.log | sed 's/^[^{]*//' > $D/${TAG}_C2_bench_line_under_rocprof.json
