# Round-5 soak on the final tree (builder records; the seeded slices of the same loops run under -m gpu)
mkdir -p gpurun_out/r05soak; OUT=gpurun_out/r05soak
python seq-align_amd/tools/fuzz_e2e.py --seconds ${1:-420} --seed ${4:-5501} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05soak/r05_fuzz_e2e.txt
tail -2 gpurun_out/r05soak/r05_fuzz_e2e.txt
python seq-align_amd/tools/x2_check.py ${2:-60} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05soak/r05_x2_check.txt
grep "x2_check:" gpurun_out/r05soak/r05_x2_check.txt
python seq-align_amd/tools/fuzz.py --seconds ${3:-240} --seed ${5:-5502} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05soak/r05_fuzz.txt
tail -2 gpurun_out/r05soak/r05_fuzz.txt
