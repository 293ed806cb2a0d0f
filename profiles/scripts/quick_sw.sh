# quick check of the multi-hit path after a kernel change: the sweep / hit-list tests, then kernel durations of C3 / C4 up to 4 hits
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q -k "sweep or sw_ or enumeration or hit or direction_byte or long_walks or soak" 2>&1 | grep -v amdgpu.ids | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in C3 C4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/quick_$wl -o t -- python $R/seq-align_amd/tools/sw_enum_profile.py $wl 4 > $R/gpurun_out/r05/quick_$wl.log 2>&1
  grep "max_hits" $R/gpurun_out/r05/quick_$wl.log | tail -1
  find $R/gpurun_out/r05/quick_$wl -name "*kernel_stats.csv" -exec head -4 {} \; | cut -c1-150
done
