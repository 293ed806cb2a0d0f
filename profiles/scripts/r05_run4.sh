mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gputests4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05/gputests4.log
tail -4 gpurun_out/r05/gputests4.log
python seq-align_amd/tools/reduce_bench.py 2>&1 | grep -v amdgpu.ids | tail -8
python seq-align_amd/tools/sw_stages.py 1 > gpurun_out/r05/sw_stages_1.txt 2>&1
python seq-align_amd/tools/sw_stages.py 4 > gpurun_out/r05/sw_stages_4.txt 2>&1
python seq-align_amd/tools/nw_stages.py > gpurun_out/r05/nw_stages.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05/sw_stages_1.txt | tail -30
bash profiles/scripts/e2e_roofline.sh r05a > gpurun_out/r05/e2e_roofline_r05a.log 2>&1
tail -5 gpurun_out/r05/e2e_roofline_r05a.log
