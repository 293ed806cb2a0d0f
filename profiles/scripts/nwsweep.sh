# seqalign_nw_batch wall clock against the number of sub-batches (option subbatches through SEQALIGN_SUBBATCHES).
# Run ON THE GPU BOX from the repo root: bash profiles/scripts/nwsweep.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
for sb in 0 1 2 3 4; do echo "C2 subbatches $sb: $(SEQALIGN_SUBBATCHES=$sb python $R/seq-align_amd/tools/nw_profile.py 10000 | tail -4 | awk '{printf "%s ", $4}')"; done
for sb in 0 1 4 8 12 16; do echo "C5share subbatches $sb: $(SEQALIGN_SUBBATCHES=$sb python $R/seq-align_amd/tools/nw_profile.py 125000 | tail -4 | awk '{printf "%s ", $4}')"; done
