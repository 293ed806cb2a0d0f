cd /tmp && export TMPDIR=/tmp
for n in 10000 125000; do
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /root/repo/gpurun_out/nwtrace_$n -- python /root/repo/seq-align_amd/tools/nw_profile.py $n > /root/repo/gpurun_out/nwtrace_$n.log 2>&1
  grep nw_batch /root/repo/gpurun_out/nwtrace_$n.log
done
for pr in 1 0; do for sb in 0 1 2 4 8; do echo "priority $pr subbatches $sb"; SEQALIGN_PIPELINE_PRIORITY=$pr SEQALIGN_SUBBATCHES=$sb python /root/repo/seq-align_amd/tools/nw_profile.py 10000 | tail -3; done; done
for sb in 0 1 4 8 32; do echo "C5share subbatches $sb"; SEQALIGN_SUBBATCHES=$sb python /root/repo/seq-align_amd/tools/nw_profile.py 125000 | tail -3; done
