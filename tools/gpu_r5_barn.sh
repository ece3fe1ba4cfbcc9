#!/bin/bash
# round 5: the generic-K one-pass front / two-pass tail: parity tests, then Barn (K = 12) and lego bench lines of the shipped build and of variant libraries
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
T=$1; shift; O=gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_backward.py tests/test_gpu_configs.py tests/test_gpu_bench_config.py tests/test_gpu_reproducible.py tests/test_gpu_trig.py -x -q > $O/tests.log 2>&1
grep -v amdgpu.ids $O/tests.log | tail -4
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for rep in 1 2; do
for V in shipped "$@"; do
  [ $V != shipped ] && cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  [ $V == shipped ] && cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
  timeout 600 python bench.py --config barn --cpu-rays 0 --steps 5 --warmup 2 --no-fp32-class-variant > $O/barn_${V}_$rep.json 2>$O/barn_$V.err
  python - <<PY
import json
try:
    d=json.load(open("$O/barn_${V}_$rep.json")); k=d["kernels"]
    print("barn %-8s %.0f rays/s %.2f ms  fwd %.2f bwd %.2f wgrad %.2f" % ("$V", d["value"], d["ms_per_step"], k["agg_forward"]["ms_per_step"], k["agg_backward"]["ms_per_step"], k["wgrad"]["ms_per_step"]))
except Exception as e: print("$V", "ERR", e, open("$O/barn_$V.err").read()[-300:])
PY
done
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
