#!/bin/bash
# round-2 dev call: tests, shipped bench, variant benches, phase trace, render-only
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider 2>&1 | tail -60 > $O/tests.log); tail -2 $O/tests.log
timeout 300 python bench.py --cpu-rays 0 --steps 10 > $O/bench_shipped.json 2>$O/bench_shipped.err
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in wpf1 wpf3; do
  cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  timeout 300 python bench.py --cpu-rays 0 --steps 10 > $O/bench_$V.json 2>/dev/null
done
cp tools/_build/trace.so pointnerf_amd/libpnerf_hip.so
timeout 300 python tools/gpu_phase_trace.py > $O/phase_trace.json 2>$O/phase_trace.err
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
timeout 300 python bench.py --cpu-rays 0 --steps 10 --render-only > $O/bench_render_only.json 2>/dev/null
timeout 300 python bench.py --cpu-rays 0 --steps 10 > $O/bench_shipped2.json 2>/dev/null
for f in shipped wpf1 wpf3 shipped2 render_only; do python - <<PY
import json
try:
    d=json.load(open("$O/bench_$f.json")); k=d["kernels"]
    print("%-12s %.0f rays/s %.2f ms  fwd %.2f bwd %.2f wgrad %.2f" % ("$f", d["value"], d["ms_per_step"], k["agg_forward"]["ms_per_step"], k.get("agg_backward",{}).get("ms_per_step",0), k.get("wgrad",{}).get("ms_per_step",0)))
except Exception as e: print("$f", "ERR", e)
PY
done
