#!/bin/bash
# Same-box A/B of the weight-gradient kernels (run ON THE GPU BOX): previous source (variant library) | register ring with the
# unconditional refill | LDS-DMA form, over the families of the BASELINE configs, fp32 and bf16 mode.
#   bash tools/ab_bww.sh [path of the variant library built from the previous kan_bwd_weight_reg.hip]
OLD=${1:-kan-vit_amd/kanvit/_ab/libkanvit_oldbww.so}
for t in cheby efficientkan fast sine; do
  for mode in "" amp; do
    if [ -f "$OLD" ]; then echo "== $t $mode old source"; KANVIT_LIB=$OLD timeout -k 10 120 python tools/time_op.py $mode $t | grep -i "weight"; fi
    echo "== $t $mode register ring"; KANVIT_BW_NO_DMA=1 timeout -k 10 120 python tools/time_op.py $mode $t | grep -i "weight"
    if [ $t = cheby ]; then echo "== $t $mode LDS-DMA"; timeout -k 10 120 python tools/time_op.py $mode $t | grep -i "weight"; fi
  done
done
