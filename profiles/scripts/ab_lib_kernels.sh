# kernel durations of two builds of the library on one box: lib/libseqalign_hip.so against lib/libseqalign_hip_exp.so (rocprofv3 --kernel-trace --stats)
WL=${1:-C4}; H=${2:-1}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05/abk
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in libseqalign_hip.so libseqalign_hip_exp.so; do
  SEQALIGN_LIB=$R/seq-align_amd/lib/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/abk/$lib.$rep -o t -- python $R/seq-align_amd/tools/sw_enum_profile.py $WL $H > $R/gpurun_out/r05/abk/$lib.$rep.log 2>&1
  echo "== $lib"; find $R/gpurun_out/r05/abk/$lib.$rep -name "*kernel_stats.csv" -exec head -4 {} \; | cut -d, -f1-4 | grep -v Name
done; done
