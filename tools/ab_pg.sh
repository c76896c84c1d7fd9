#!/bin/bash
# Same-box A/B of the gathering patch-embedding weight gradient (run ON THE GPU BOX): bench lines of two workloads per library.
for lib in "" "$@"; do
  for w in vitb16-224-sine+fourier vitb16-224-cheby; do
    echo "== [$lib] $w"
    KANVIT_LIB=$lib python bench.py --workload $w --no-cpu-baseline --no-amp-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   ms/step', d['ms_per_step'])
for k,v in d.get('kernels',{}).items():
    if k.startswith('layer') or 'weight' in k: print('    ',k, v['avg_ms'])
"
  done
done
