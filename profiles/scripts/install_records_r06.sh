# HERE (not on the GPU box): copy what `r06_records.sh <tag>` left under gpurun_out/ into profiles/r06/ (the tracked records)
TAG=${1:?tag}; D=profiles/r06; mkdir -p $D
cp gpurun_out/$TAG/${TAG}_bench_*.json gpurun_out/$TAG/${TAG}_gputests.log $D/ 2>/dev/null
for w in C2 C3 C4 C5; do cp gpurun_out/$TAG/${TAG}_$w.json gpurun_out/$TAG/${TAG}_${w}_kernel_stats.csv gpurun_out/$TAG/${TAG}_${w}_bench_line_under_rocprof.json $D/ 2>/dev/null; done
if [ -d gpurun_out/e2e_roofline_$TAG ]; then
  for f in gpurun_out/e2e_roofline_$TAG/*_kernel_stats.csv; do cp $f $D/${TAG}_e2e_$(basename $f); done
  cp gpurun_out/e2e_roofline_$TAG/e2e_pmc_insts.json $D/${TAG}_e2e_pmc_insts.json
  cp gpurun_out/e2e_roofline_$TAG/e2e_roofline.json $D/${TAG}_e2e_roofline.json
  cp gpurun_out/e2e_roofline_$TAG/e2e_roofline.json profiles/e2e_roofline.json
fi
ls $D | grep "^$TAG" | wc -l
