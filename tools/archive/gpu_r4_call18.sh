#!/bin/bash
# round 4, call 18: fused colour loss: parity test, then A/B (--unfused-color-loss) on the four configurations' headline one
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c18; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_steps.py tests/test_gpu_bench_dist.py -q > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -4
for i in 1 2; do
timeout 300 python bench.py --cpu-rays 0 --steps 20 --no-fp32-class-variant --unfused-color-loss > $O/bench_unfused_$i.json 2>/dev/null
timeout 300 python bench.py --cpu-rays 0 --steps 20 --no-fp32-class-variant > $O/bench_fused_$i.json 2>/dev/null
done
timeout 300 python bench.py --config chair --cpu-rays 0 --steps 20 > $O/bench_chair.json 2>/dev/null
timeout 300 python bench.py --config chair --cpu-rays 0 --steps 20 --unfused-color-loss > $O/bench_chair_unfused.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4c18/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"]), "rays/s", round(d["ms_per_step"],3), "ms  outside library kernels", round(d.get("ms_outside_library_kernels",0),3), "loss", d["config"].get("final_loss"))
    except Exception as e: print(f, "ERR", e)
PY
