#!/bin/bash
# Static VALU mix of the hot loops of the VALU-bound kernels on the host-level paths (run anywhere hipcc is: no GPU needed):
#   bash profiles/scripts/valu_mix_table.sh > profiles/r04/r04_valu_mix.json
# One line of JSON per kernel instantiation (matched by name and the first two template arguments -- the third, the LDS bytes per pair,
# changed when the direction bytes went into blocks); profiles/scripts/e2e_roofline.sh looks kernels up here by demangled name.
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$R/seq-align_amd/tools/valu_mix.py
C=$R/seq-align_amd/csrc
echo "["
first=1
emit() { # file function mangled-args demangled [second needle of the mangled name] [what the demangled name ends with]
  out=$(python "$T" "$C/$1" "$2" "$3" $5 2>/dev/null | tail -1)
  [ -z "$out" ] && return
  [ $first = 1 ] || echo ","
  first=0
  echo "{\"demangled\": \"$4\", \"ends\": \"$6\", \"mix\": $out}"
}
# (round 6: the NW / best-hit fills exist in two forms -- the last template argument: the direction byte's LOCAL form, sa_kernels.h)
for form in "ELb1E|, true>" "ELb0E|, false>"; do
  nd=${form%%|*}; en=${form##*|}
  emit sa_fill_dirs_x2.hip fill_nw_dirs_x2_kernel ILi3ELi0ELi "fill_nw_dirs_x2_kernel<3, 0," $nd "$en"
  emit sa_fill_dirs_x2.hip fill_nw_dirs_x4_kernel ILi5ELi0ELi "fill_nw_dirs_x4_kernel<5, 0," $nd "$en"
  emit sa_fill_dirs_x2.hip fill_nw_dirs_x4x2_kernel ILi5ELi0ELi "fill_nw_dirs_x4x2_kernel<5, 0," $nd "$en"
  emit sa_fill_dirs_x2.hip fill_sw_best_x4_kernel ILi5ELi0ELi "fill_sw_best_x4_kernel<5, 0," $nd "$en"
  emit sa_fill_dirs_x2.hip fill_sw_best_x4x2_kernel ILi5ELi0ELi "fill_sw_best_x4x2_kernel<5, 0," $nd "$en"
  emit sa_fill_dirs_x2.hip fill_sw_best_x2_kernel ILi3ELi0ELi "fill_sw_best_x2_kernel<3, 0," $nd "$en"
  emit sa_fill_dirs_x2.hip fill_sw_best_x2_kernel ILi5ELi1ELi "fill_sw_best_x2_kernel<5, 1," $nd "$en"
  emit sa_fill_dirs.hip fill_nw_dirs_kernel ILi3ELi0ELi "fill_nw_dirs_kernel<3, 0," $nd "$en"
done
emit sa_fill_dirs_x2.hip fill_dirs_x2_kernel ILi3ELi0ELi "fill_dirs_x2_kernel<3, 0,"
emit sa_fill_dirs_x2.hip fill_dirs_x2_kernel ILi5ELi1ELi "fill_dirs_x2_kernel<5, 1,"
emit sa_sw_sweep.hip sw_sweep_dirs_kernel ILi3E "sw_sweep_dirs_kernel<3"
emit sa_sw_sweep.hip sw_sweep_dirs_kernel ILi5E "sw_sweep_dirs_kernel<5"
emit sa_sw_sweep.hip sw_sweep_dirs_ev_kernel ILi3Ej "sw_sweep_dirs_ev_kernel<3"
emit sa_sw_sweep.hip sw_sweep_dirs_ev_kernel ILi5Ej "sw_sweep_dirs_ev_kernel<5"
echo "]"
