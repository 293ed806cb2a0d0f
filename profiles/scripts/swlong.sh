# seqalign_sw_batch on wide and on few long pairs (tools/sw_wide_bench.py, tools/sw_long_bench.py): wall clock + per-kernel times.
# Run ON THE GPU BOX from the repo root: bash profiles/scripts/swlong.sh  -> gpurun_out/swlong_{wide,long}_kernel_stats.csv, gpurun_out/swlong.log
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
: > $R/gpurun_out/swlong.log
for w in wide long; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/swlong_$w -o s -- python $R/seq-align_amd/tools/sw_${w}_bench.py 2>/dev/null | grep "hits" >> $R/gpurun_out/swlong.log
  find $R/gpurun_out/swlong_$w -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/swlong_${w}_kernel_stats.csv \;
done
cat $R/gpurun_out/swlong.log
