#!/bin/bash
# dev: bench the shipped library and the variant libraries given as arguments (tools/_build/<name>.so) back to back on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in shipped "$@"; do
  [ $V != shipped ] && cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  PNERF_BENCH_ALLOW_NAN=1 timeout 300 python bench.py --cpu-rays 0 --steps 10 --no-variants > $O/bench_$V.json 2>$O/bench_$V.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$V.json")); k=d["kernels"]
    print("%-12s %.0f rays/s %.2f ms  fwd %.2f bwd %.2f wgrad %.2f color %.2f+%.2f" % ("$V", d["value"], d["ms_per_step"], k["agg_forward"]["ms_per_step"], k["agg_backward"]["ms_per_step"], k["wgrad"]["ms_per_step"], k["color_forward"]["ms_per_step"], k["color_backward"]["ms_per_step"]))
except Exception as e: print("$V", "ERR", e, open("$O/bench_$V.err").read()[-300:])
PY
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
