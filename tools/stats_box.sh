#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel-trace stats only (no PMC passes) of one bench.py workload -> gpurun_out/stats/<name>.md
#   bash tools/stats_box.sh <name> [bench.py args...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/stats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-4}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $NAME -- python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-kernel-timer --no-amp-leg --graph off "$@" > $OUT/$NAME.log 2>&1 || { tail -5 $OUT/$NAME.log; exit 1; }
python3 $R/tools/summarize_prof.py stats $OUT/${NAME}_kernel_stats.csv $((STEPS + 1)) > $OUT/$NAME.md
rm -f $OUT/${NAME}_kernel_trace.csv $OUT/${NAME}_agent_info.csv $OUT/${NAME}_domain_stats.csv
