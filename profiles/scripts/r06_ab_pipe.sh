# Round 6: the packed fills' LDS traffic a row ahead (SA_X2_LDS_PIPE / SA_X2_FLUSH_DEFER, csrc/sa_fill_dirs_x2.hip) -- correctness, then a
# same-box A/B of the kernels' durations: lib/libseqalign_hip.so (both on) against lib/libseqalign_hip_exp_off.so (round 5's form)
# and lib/libseqalign_hip_exp_prof.so (the profile pipeline without the deferred flush).  On the GPU box, from the repo root:
#     bash profiles/scripts/r06_ab_pipe.sh [check] [ab] [size]
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_ab; mkdir -p $O
STAGES=${*:-check ab size}
LIBS="libseqalign_hip.so libseqalign_hip_exp_off.so libseqalign_hip_exp_prof.so"
for s in $STAGES; do
case $s in
check)
  cd $R
  timeout 600 python seq-align_amd/tools/x2_check.py 60 2>&1 | grep -v amdgpu.ids | tail -6
  timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_soak.py tests/test_gpu_cigar.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sw_batch or nw_batch or refwalk or reference_walked" 2>&1 | grep -v amdgpu.ids | tail -4 ;;
ab)
  cd /tmp && export TMPDIR=/tmp
  for rep in 1 2; do
  for lib in $LIBS; do
    [ -f $R/seq-align_amd/lib/$lib ] || continue
    for w in "sw C4 1" "sw C4 4" "sw C3 1" "sw C3 4" "nw 10000" "nw 125000"; do
      set -- $w
      if [ $1 = sw ]; then cmd="python $R/seq-align_amd/tools/sw_enum_profile.py $2 $3"; key=$2_$3; else cmd="python $R/seq-align_amd/tools/nw_profile.py $2"; key=nw_$2; fi
      SEQALIGN_LIB=$R/seq-align_amd/lib/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$lib.$key.$rep -o t -- $cmd > $O/$lib.$key.$rep.log 2>&1
      echo "== rep $rep $lib $key: $(grep -h ' ms' $O/$lib.$key.$rep.log | tail -1)"
      find $O/$lib.$key.$rep -name "*kernel_stats.csv" -exec head -4 {} \; | cut -d, -f1-4 | grep "fill_\|sweep\|traceback" | sed 's/^/      /'
    done
  done; done ;;
size)
  cd $R
  for lib in $LIBS; do
    [ -f $R/seq-align_amd/lib/$lib ] || continue
    echo "== $lib"; SEQALIGN_LIB=$R/seq-align_amd/lib/$lib bash profiles/scripts/c4_by_batch_size.sh 2>&1 | grep -v "^pairs" | sed 's/^/      /'
  done ;;
esac
done
