#!/bin/bash
# (see tools/kstat.sh for how to make file.s)
# kloop.sh file.s pattern : instruction mix between the first and the last MFMA of each matching kernel (the unrolled main loop body)
f=$1; pat=$2
grep -n "^_ZN[^ ]*: " $f | grep -E "$pat" | while IFS=: read a name rest; do
  b=$(awk -v a=$a 'NR>a && /s_endpgm/ {print NR; exit}' $f)
  sed -n "${a},${b}p" $f > /tmp/_kanvit_body.s
  f1=$(grep -n 'v_mfma' /tmp/_kanvit_body.s | head -1 | cut -d: -f1); l1=$(grep -n 'v_mfma' /tmp/_kanvit_body.s | tail -1 | cut -d: -f1)
  sed -n "${f1},${l1}p" /tmp/_kanvit_body.s > /tmp/_kanvit_loop.s
  echo "$(echo $name | c++filt | cut -c28-110)  mfma-span: lines $((l1-f1)) valu $(grep -c '^\s*v_' /tmp/_kanvit_loop.s) mfma $(grep -c 'v_mfma' /tmp/_kanvit_loop.s) br $(grep -c 's_cbranch' /tmp/_kanvit_loop.s) trans $(grep -c 'v_exp_f32\|v_rcp_f32\|v_log_f32\|v_sqrt\|v_rsq' /tmp/_kanvit_loop.s) cndmask $(grep -c 'v_cndmask' /tmp/_kanvit_loop.s) mov $(grep -c 'v_mov_b32\|v_accvgpr' /tmp/_kanvit_loop.s) ds $(grep -c '^\s*ds_' /tmp/_kanvit_loop.s) vmem $(grep -c '^\s*global_\|^\s*buffer_' /tmp/_kanvit_loop.s)"
done
