#!/bin/bash
# dev: timing ablations of k_agg_backward
for m in 0 1 2 4 7; do
  echo "== PNERF_DEBUG_SKIP=$m"
  PNERF_DEBUG_SKIP=$m python bench.py --steps 3 --warmup 1 --rays 65536 --cpu-rays 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print('step ms %.1f' % d['ms_per_step'], {n: round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>0.5})"
done
