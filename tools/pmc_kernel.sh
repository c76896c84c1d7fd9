#!/bin/bash
# Run ON THE GPU BOX: PMC passes (4-5 counters each, never mixed with trace domains) over tools/bench_op.py and print
# per-dispatch averages for the kernels whose name matches $1.   usage: pmc_kernel.sh <kernel regex> <bench_op args...>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
PAT=$1; shift
OUT=$R/gpurun_out/pmck
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CS in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
          "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
          "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
          "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
          "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
          "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM" \
          "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
          "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $CS --output-format csv -d $OUT -o p$i -- python3 $R/tools/bench_op.py "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
python3 - "$PAT" $OUT <<'PY'
import csv, glob, re, sys
from collections import defaultdict
pat = re.compile(sys.argv[1]); out = sys.argv[2]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + "/p*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k)
        if pat.search(k):
            a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"   {c:32s} {v / n:16.1f}   (n={n})")
PY
rm -f $OUT/*_counter_collection.csv
