#!/bin/bash
# round 6: the mixed-format GEMMs on the device: tile GEMM tests, the parity tests under the candidate defaults, bench A/B
cd $GRAFT_REPO_ROOT
T=${1:-r6_mix}; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests/test_gpu_mix.py -q -s -m gpu > $O/test_mix.log 2>&1; echo "mix rc $?"; tail -3 $O/test_mix.log
for M in 5 7; do
  PNERF_MIX_MASK=$M timeout 1500 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_render.py tests/test_gpu_backward.py tests/test_gpu_trig.py tests/test_gpu_train_steps.py tests/test_gpu_configs.py -q -s -m gpu > $O/tests_mask$M.log 2>&1
  echo "mask $M rc $?"; tail -4 $O/tests_mask$M.log
done
for M in 7 5 0; do
  PNERF_MIX_MASK=$M python bench.py --steps 10 --warmup 4 > $O/bench_mask$M.json 2> $O/bench_mask$M.err
  python - <<P
import json
d=json.load(open("$O/bench_mask$M.json"))
k=d.get("kernels") or d.get("kernel_ms") or {}
print("mask $M", d["value"], d["ms_per_step"], {n:round(v["ms_per_step"],2) for n,v in k.items() if isinstance(v,dict) and v.get("ms_per_step",0)>0.3})
P
done
