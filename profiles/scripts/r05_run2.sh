mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q -k "sweep or sw_ or soak or pool or enumeration or hit" > gpurun_out/r05/gputests2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05/gputests2.log
tail -5 gpurun_out/r05/gputests2.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in C3 C4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/sweep2_$wl -o t -- python $R/seq-align_amd/tools/sw_enum_profile.py $wl 4 > $R/gpurun_out/r05/sweep2_$wl.log 2>&1
  grep -v amdgpu.ids $R/gpurun_out/r05/sweep2_$wl.log | grep "max_hits"
  find $R/gpurun_out/r05/sweep2_$wl -name "*kernel_stats.csv" -exec head -6 {} \;
done
