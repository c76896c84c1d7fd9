#!/bin/bash
# GPU box: one short bench per workload (no CPU baseline), compact summary.
for w in vitb16-224-efficientkan vits16-224-fast vitb16-224-sine vitb16-224-fourier vits16-224-cheby cifar-cheby-default; do
  timeout -k 10 280 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/bench_$w.err > gpurun_out/bench_$w.json
  echo "== $w rc=$?"
  python - "$w" <<'PY'
import json,sys
try:
    d=json.load(open(f"gpurun_out/bench_{sys.argv[1]}.json"))
    print(d["value"], "img/s", d["ms_per_step"], "ms/step; custom", d.get("custom_kernel_ms_per_step"))
    for k,v in d["kernels"].items(): print("   ", k, v["avg_ms"], "ms", v["TFLOP/s"], "TF/s", v["launches_per_step"])
except Exception as e:
    print("FAILED", e); print(open(f"gpurun_out/bench_{sys.argv[1]}.err").read()[-1500:])
PY
done
