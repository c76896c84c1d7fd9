#!/bin/bash
# Build an A/B variant of libkanvit.so: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DKV_EXPERIMENT=1]
#   -> kan-vit_amd/kanvit/_ab/libkanvit_<name>.so   (run with KANVIT_LIB=<that path>; git-ignored, travels with gpurun)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
O=$R/kan-vit_amd/kanvit/_ab; mkdir -p $O/obj_$N
cd $R/kan-vit_amd/csrc
pids=()
# sources and flags come from kanvit/build.py: the two builds cannot drift (tests/test_abi_cpu.py checks this file has no list of its own)
SRCS=$(python3 -c "import sys; sys.path.insert(0, '$R/kan-vit_amd/kanvit'); import build; print(' '.join(s[:-4] for s in build.SOURCES))")
FLAGS=$(python3 -c "import sys; sys.path.insert(0, '$R/kan-vit_amd/kanvit'); import build; print(' '.join(build.FLAGS))")
for f in $SRCS; do
  hipcc $FLAGS -I ../../include "$@" -c $f.hip -o $O/obj_$N/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $O/obj_$N/*.o -o $O/libkanvit_$N.so
rm -rf $O/obj_$N
echo $O/libkanvit_$N.so
