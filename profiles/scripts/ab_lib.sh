# same-box A/B of two builds of the library: lib/libseqalign_hip.so against lib/libseqalign_hip_exp.so (make exp EXPFLAGS=...)
H=${1:-1}
for r in 1 2 3; do
for lib in libseqalign_hip.so libseqalign_hip_exp.so; do
echo "== $lib"; SEQALIGN_LIB=$PWD/seq-align_amd/lib/$lib python seq-align_amd/tools/sw_stages.py $H 2>&1 | grep "wall" | tail -5 | tr '\n' ' '; echo
done; done
