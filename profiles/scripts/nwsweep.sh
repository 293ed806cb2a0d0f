for sb in 0 1 2 4; do echo "C2 subbatches $sb"; SEQALIGN_SUBBATCHES=$sb python /root/repo/seq-align_amd/tools/nw_profile.py 10000 | tail -3; done
for sb in 0 1 4 8 16; do echo "C5share subbatches $sb"; SEQALIGN_SUBBATCHES=$sb python /root/repo/seq-align_amd/tools/nw_profile.py 125000 | tail -3; done
