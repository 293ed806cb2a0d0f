set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -x -q -k "walker or long_walks or reports_its_kernels" 2>&1 | grep -v amdgpu.ids | tail -6
python seq-align_amd/tools/walk_group_ab.py C2 C3 C4 2>&1 | grep -v amdgpu.ids
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_walk -o t -- python $GRAFT_REPO_ROOT/seq-align_amd/tools/walk_group_ab.py C2 C4 > $GRAFT_REPO_ROOT/gpurun_out/r06_walk.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r06_walk -name "*kernel_stats.csv" -exec grep -h "traceback" {} \; | cut -c1-150
cd $GRAFT_REPO_ROOT
bash profiles/scripts/r06_pmc_fills.sh
