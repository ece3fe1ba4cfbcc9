#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
timeout 300 python bench.py --cpu-rays 0 --steps 10 > $O/bench_shipped.json 2>$O/bench_shipped.err
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in plain; do
  cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  timeout 300 python bench.py --cpu-rays 0 --steps 10 > $O/bench_$V.json 2>/dev/null
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
for f in shipped plain; do python - <<PY
import json
try:
    d=json.load(open("$O/bench_$f.json")); k=d["kernels"]
    print("%-12s %.0f rays/s %.2f ms  fwd %.2f bwd %.2f wgrad %.2f" % ("$f", d["value"], d["ms_per_step"], k["agg_forward"]["ms_per_step"], k.get("agg_backward",{}).get("ms_per_step",0), k.get("wgrad",{}).get("ms_per_step",0)))
except Exception as e: print("$f", "ERR", e)
PY
done
bash tools/gpu_pmc_mfma.sh r2h
