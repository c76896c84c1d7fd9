#!/bin/bash
# Instruction mix of whole kernels in a device assembly dump:
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -fno-finite-math-only -I include -S --cuda-device-only kan-vit_amd/csrc/kan_fwd_reg.hip -o /tmp/k.s
#   bash tools/kstat.sh /tmp/k.s 'kan_fwd_reg_kernelILi3E'      (egrep pattern on the MANGLED name)
f=$1; pat=$2
grep -n "^_ZN[^ ]*: " $f | grep -E "$pat" | while IFS=: read a name rest; do
  b=$(awk -v a=$a 'NR>a && /s_endpgm/ {print NR; exit}' $f)
  sed -n "${a},${b}p" $f > /tmp/_kanvit_body.s
  echo "$(echo $name | c++filt | cut -c1-120)"
  echo "   lines $((b-a)) valu $(grep -c '^\s*v_' /tmp/_kanvit_body.s) mfma $(grep -c 'v_mfma' /tmp/_kanvit_body.s) branches $(grep -c 's_cbranch' /tmp/_kanvit_body.s) trans $(grep -c 'v_exp_f32\|v_rcp_f32\|v_log_f32\|v_sqrt\|v_rsq' /tmp/_kanvit_body.s) div $(grep -c 'v_div_' /tmp/_kanvit_body.s) scratch $(grep -c 'scratch_' /tmp/_kanvit_body.s)"
done
