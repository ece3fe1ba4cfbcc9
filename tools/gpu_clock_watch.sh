#!/bin/bash
# dev: sample clock / power with rocm-smi while bench.py runs (shipped library, then the variants given): is the step power-limited?
cd $GRAFT_REPO_ROOT
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in shipped "$@"; do
  [ $V != shipped ] && cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  ( PNERF_BENCH_ALLOW_NAN=1 timeout 300 python bench.py --cpu-rays 0 --steps 150 --warmup 3 > gpurun_out/clk_$V.json 2>/dev/null ) &
  BP=$!
  while kill -0 $BP 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.4
  done | awk '{gsub(/[()Mhz]/,"",$2); if ($2+0 > 400) print}' | sort | uniq -c | sort -rn | head -8
  wait $BP
  python -c "
import json; d=json.load(open('gpurun_out/clk_$V.json')); k=d['kernels']; print('$V', round(d['ms_per_step'],2), 'fwd', round(k['agg_forward']['ms_per_step'],2), 'bwd', round(k['agg_backward']['ms_per_step'],2), 'wgrad', round(k['wgrad']['ms_per_step'],2))"
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
