# does waking the host pool's sleeping workers show in the calls?  SEQALIGN_HOST_SPIN_US = 400 (default) vs 3000
for r in 1 2 3; do
for us in 400 3000; do
echo "== spin $us us"
SEQALIGN_HOST_SPIN_US=$us python seq-align_amd/tools/sw_stages.py 1 2>&1 | grep "wall\|expanded" | tail -4 | tr '\n' ' '; echo
SEQALIGN_HOST_SPIN_US=$us python seq-align_amd/tools/sw_stages.py 4 2>&1 | grep "wall" | tail -5 | tr '\n' ' '; echo
SEQALIGN_HOST_SPIN_US=$us python seq-align_amd/tools/nw_stages.py 2>&1 | grep "wall" | tr '\n' ' '; echo
done; done
