#!/bin/bash
# round 4, call 23: the zero-one regulariser inside the render node: parity, then A/B against its own pass (PNERF_ZERO_ONE_IN_RENDER=0)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_steps.py tests/test_gpu_bench_dist.py tests/test_gpu_model_shell.py tests/test_gpu_training_loop.py -q > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -4
for i in 1 2; do
PNERF_ZERO_ONE_IN_RENDER=0 timeout 300 python bench.py --cpu-rays 0 --steps 20 --no-fp32-class-variant > $O/bench_ownpass_$i.json 2>/dev/null
timeout 300 python bench.py --cpu-rays 0 --steps 20 --no-fp32-class-variant > $O/bench_inrender_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4c23/bench_*.json")):
    try:
        d=json.load(open(f)); k=d["kernels"]; print(f.split("/")[-1], round(d["value"]), "rays/s", round(d["ms_per_step"],3), "ms  outside", round(d.get("ms_outside_library_kernels",0),3), "gather", round(k["gather"]["ms_per_step"],3), "bwd", round(k["agg_backward"]["ms_per_step"],3), "loss", d["config"].get("final_loss"))
    except Exception as e: print(f, "ERR", e)
PY
