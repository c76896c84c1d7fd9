#!/bin/bash
# GPU box: one bench per workload (no CPU baseline); the default line carries the fp32 value and the amp_bf16 / ff_bf16x3 legs.
for w in vitb16-224-efficientkan vits16-224-fast vitb16-224-sine vitb16-224-fourier vits16-224-cheby cifar-cheby-default mnist-cheby-tiny; do
  timeout -k 10 280 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench_$w.err > gpurun_out/bench_$w.json
  python - "$w" <<'PY'
import json,sys
try:
    d=json.load(open(f"gpurun_out/bench_{sys.argv[1]}.json"))
    print(f"{sys.argv[1]:26s} fp32 {d['value']:9.1f} img/s {d['ms_per_step']:8.2f} ms | amp {d.get('amp_bf16',{}).get('value','-')} | ff_bf16x3 {d.get('ff_bf16x3',{}).get('value','-')} | graph {d['config']['hip_graph']}")
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(f"gpurun_out/bench_{sys.argv[1]}.err").read()[-800:])
PY
done
