# the command-line tool after a change to its reader / printer: its tests, the 1 M-pair record (tools/cli_bench.py), and -- when a
# previous build of the tool lies beside it as bin/seqalign_nw_prev -- the two outputs compared byte for byte
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q -k "cli" 2>&1 | grep -v amdgpu.ids | tail -3
python seq-align_amd/tools/cli_bench.py 1000000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/cli_bench_new.txt
if [ -x seq-align_amd/bin/seqalign_nw_prev ]; then
  cp /tmp/cli_bench/out.txt /tmp/cli_bench/out_new.txt
  for i in 1 2; do
    python3 -c 'import subprocess,sys,time; t=time.perf_counter(); subprocess.run([sys.argv[1],"--printscores","--file","/tmp/cli_bench/c5_1000000.fa"],stdout=open("/tmp/cli_bench/out_prev.txt","wb"),check=True); print("previous build: %.3f s wall" % (time.perf_counter()-t))' seq-align_amd/bin/seqalign_nw_prev
  done 2>&1 | tee -a gpurun_out/r05/cli_bench_new.txt
  cmp /tmp/cli_bench/out_new.txt /tmp/cli_bench/out_prev.txt && echo "outputs identical ($(stat -c %s /tmp/cli_bench/out_new.txt) bytes)" | tee -a gpurun_out/r05/cli_bench_new.txt
fi
for i in 1 2 3; do
  SEQALIGN_CLI_EXIT=full python3 -c 'import subprocess,sys,time; t=time.perf_counter(); subprocess.run([sys.argv[1],"--printscores","--file","/tmp/cli_bench/c5_1000000.fa"],stdout=open("/tmp/cli_bench/out_full.txt","wb"),check=True); print("SEQALIGN_CLI_EXIT=full: %.3f s wall" % (time.perf_counter()-t))' seq-align_amd/bin/seqalign_nw
done 2>&1 | tee -a gpurun_out/r05/cli_bench_new.txt
cmp /tmp/cli_bench/out_new.txt /tmp/cli_bench/out_full.txt && echo "full exit: output identical" | tee -a gpurun_out/r05/cli_bench_new.txt
