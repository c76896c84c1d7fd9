#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats of the default bench, then two separate
# PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; --pmc is never combined
# with other trace domains).  Output lands in gpurun_out/prof/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-3}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-kernel-timer --no-amp-leg "$@" > $OUT/trace.log 2>&1 || exit 1
python3 $R/tools/summarize_prof.py stats $OUT/trace_kernel_stats.csv $((STEPS + 1)) > $OUT/kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT -o pmc_$C -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-amp-leg "$@" > $OUT/pmc_$C.log 2>&1 || exit 1
  python3 $R/tools/summarize_prof.py pmc $OUT/pmc_${C}_counter_collection.csv $C > $OUT/pmc_$C.md
done
SUF=""; case " $* " in *" bf16 "*) SUF="_bf16";; esac
python3 $R/tools/pmc_to_traffic.py $OUT/pmc_FETCH_SIZE_counter_collection.csv $OUT/pmc_WRITE_SIZE_counter_collection.csv "$SUF" > $OUT/traffic.json
rm -f $OUT/trace_kernel_trace.csv $OUT/pmc_*_counter_collection.csv
ls -la $OUT
