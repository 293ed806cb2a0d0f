# where seqalign_sw's time goes on 200 000 read / window pairs with the tool's default hit cap: the library's stage laps (option timing)
mkdir -p gpurun_out/r05
python seq-align_amd/tools/cli_bench.py 10000 200000 2>&1 | grep -v amdgpu.ids | grep "seqalign_sw\|stages"
for i in 1 2; do
  SEQALIGN_TIMING=1 SEQALIGN_CLI_TIMING=1 seq-align_amd/bin/seqalign_sw --file /tmp/cli_bench/c3_200000.fa > /tmp/cli_bench/o.txt 2> gpurun_out/r05/cli_sw_stages_$i.txt
  grep -v amdgpu.ids gpurun_out/r05/cli_sw_stages_$i.txt | tail -40
done
