#!/bin/bash
# Run ON THE GPU BOX (gpurun): everything the committed profiles/rNN_* files of a round are made from.
#   bash tools/round_profiles.sh <tag>      -> gpurun_out/<tag>/...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
STAGE=${2:-all}          # 1 = kernel stats + FETCH/WRITE passes, 2 = SQ counter passes, 3 = bench lines (a gpurun call is at most 20 min)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$STAGE" = all ] || [ "$STAGE" = 1 ]; then
# 1. kernel-trace stats + FETCH/WRITE PMC passes of the headline workload, fp32 and bf16 autocast
bash tools/profile_box.sh > $O/profile_fp32.log 2>&1 || { echo "profile fp32 failed"; tail -5 $O/profile_fp32.log; exit 1; }
for f in kernel_stats.md pmc_FETCH_SIZE.md pmc_WRITE_SIZE.md traffic.json; do cp gpurun_out/prof/$f $O/vitb16_cheby_$f; done
bash tools/profile_box.sh --amp bf16 > $O/profile_bf16.log 2>&1 || { echo "profile bf16 failed"; tail -5 $O/profile_bf16.log; exit 1; }
for f in kernel_stats.md pmc_FETCH_SIZE.md pmc_WRITE_SIZE.md traffic.json; do cp gpurun_out/prof/$f $O/vitb16_cheby_amp_bf16_$f; done
echo "profiles done"
fi
if [ "$STAGE" = all ] || [ "$STAGE" = 2 ]; then
# 2. SQ counters of the custom kernels (one MSA block, forward + backward), fp32 and bf16
bash tools/pmc_kernel.sh "kan_|attn" bwd fp32 3 > $O/sq_pmc_fp32.txt 2>&1 || { echo "sq fp32 failed"; tail -5 $O/sq_pmc_fp32.txt; exit 1; }
bash tools/pmc_kernel.sh "kan_|attn" bwd amp 3 > $O/sq_pmc_bf16.txt 2>&1 || { echo "sq bf16 failed"; tail -5 $O/sq_pmc_bf16.txt; exit 1; }
bash tools/pmc_kernel.sh "kan_" bwd fp32 3 efficientkan > $O/sq_pmc_efficientkan_fp32.txt 2>&1 || { echo "sq efficientkan failed"; exit 1; }
bash tools/pmc_kernel.sh "kan_" bwd amp 3 fast vits > $O/sq_pmc_fast_vits_bf16.txt 2>&1 || { echo "sq fast failed"; exit 1; }
echo "sq done"
fi
if [ "$STAGE" = all ] || [ "$STAGE" = 3 ]; then
# 3. bench lines
python bench.py > $O/bench_vitb16_cheby.json 2> $O/bench_vitb16_cheby.err || exit 1
python bench.py --amp bf16 --no-cpu-baseline > $O/bench_vitb16_cheby_amp_bf16.json 2>> $O/bench.err || exit 1
python bench.py --workload mnist-cheby-tiny --steps 300 --warmup 20 > $O/bench_mnist_cheby_tiny.json 2>> $O/bench.err || exit 1
python bench.py --workload cifar-cheby-default --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_cifar_cheby_default.json 2>> $O/bench.err || exit 1
# rocprofv3 kernel-trace stats of the other fp32 workloads DESIGN.md quotes (event-timed only in round 2)
for W in vitb16-224-efficientkan vits16-224-fast vitb16-224-sine+fourier; do
  T=$(echo $W | tr '+' '_' | tr '-' '_')
  bash tools/stats_box.sh $T --workload $W > /dev/null 2>&1 && cp gpurun_out/stats/$T.md $O/${T}_kernel_stats.md
done
python bench.py --workload vits16-224-fast --amp bf16 --no-cpu-baseline > $O/bench_vits16_fast_amp_bf16.json 2>> $O/bench.err || exit 1
python bench.py --workload vits16-224-fast --no-cpu-baseline --no-amp-leg > $O/bench_vits16_fast.json 2>> $O/bench.err || exit 1
python bench.py --workload vitb16-224-efficientkan --no-cpu-baseline > $O/bench_vitb16_efficientkan.json 2>> $O/bench.err || exit 1
python bench.py --workload vitb16-224-sine+fourier --no-cpu-baseline > $O/bench_vitb16_sine_fourier.json 2>> $O/bench.err || exit 1
python bench.py --workload vitb16-224-sine+fourier --amp bf16 --no-cpu-baseline > $O/bench_vitb16_sine_fourier_amp_bf16.json 2>> $O/bench.err || exit 1
echo "bench done"
fi
ls $O
