#!/bin/bash
# Build an A/B variant of libkanvit.so: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DKV_EXPERIMENT=1]
#   -> kan-vit_amd/kanvit/_ab/libkanvit_<name>.so   (run with KANVIT_LIB=<that path>; git-ignored, travels with gpurun)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
O=$R/kan-vit_amd/kanvit/_ab; mkdir -p $O/obj_$N
cd $R/kan-vit_amd/csrc
pids=()
for f in kan_tile kan_fwd_reg kan_fwd_reg_bf16 kan_bwd_input_reg kan_bwd_input_reg_bf16 kan_bwd_weight_reg kan_layer attention addln split3 ff_small ff_epilogue kan_tiny; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -fno-finite-math-only -fvisibility=hidden -I ../../include "$@" -c $f.hip -o $O/obj_$N/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $O/obj_$N/*.o -o $O/libkanvit_$N.so
rm -rf $O/obj_$N
echo $O/libkanvit_$N.so
