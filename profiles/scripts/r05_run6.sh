mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_soak.py tests/test_gpu_multirank.py -x -q > gpurun_out/r05/gputests6.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05/gputests6.log
grep -v amdgpu.ids gpurun_out/r05/gputests6.log | tail -6
bash profiles/scripts/sweep_pmc.sh r05b C3 > gpurun_out/r05/sweep_pmc_C3_after.log 2>&1
bash profiles/scripts/sweep_pmc.sh r05b C4 > gpurun_out/r05/sweep_pmc_C4_after.log 2>&1
python profiles/scripts/pmc_db_summary.py gpurun_out/sweep_pmc_r05b_C3 > gpurun_out/sweep_pmc_r05b_C3/summary.json
python profiles/scripts/pmc_db_summary.py gpurun_out/sweep_pmc_r05b_C4 > gpurun_out/sweep_pmc_r05b_C4/summary.json
head -8 gpurun_out/sweep_pmc_r05b_C3/kernel_stats.csv
