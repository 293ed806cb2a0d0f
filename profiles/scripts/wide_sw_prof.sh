# kernel durations of the wide-row SW calls (10 000 reads of 700 bp against 1 000 bp windows): up to 4 hits, best hit
mkdir -p gpurun_out/r05
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mh in 4 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/wide_prof_$mh -o t -- python $R/seq-align_amd/tools/sw_wide_profile.py 700 $mh > $R/gpurun_out/r05/wide_prof_$mh.log 2>&1
  grep launched $R/gpurun_out/r05/wide_prof_$mh.log
  find $R/gpurun_out/r05/wide_prof_$mh -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r05/wide_sw_700_${mh}_kernel_stats.csv \;
  head -6 $R/gpurun_out/r05/wide_sw_700_${mh}_kernel_stats.csv | cut -c1-160
done
