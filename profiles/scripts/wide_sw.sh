# the wide-row SW paths after a change: their tests, then the same-process A/B of tools/sw_wide_reads.py (best hit, up to 4 hits)
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q -k "1024 or wide or sweep or direction_byte or long_walks" 2>&1 | grep -v amdgpu.ids | tail -6
for mh in 4 1; do
  python seq-align_amd/tools/sw_wide_reads.py 700 $mh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/wide_sw_700_$mh.txt
done
python seq-align_amd/tools/sw_wide_reads.py 600 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/wide_sw_600_4.txt
