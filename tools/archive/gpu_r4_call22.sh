#!/bin/bash
# round 4, call 22: colour forward with interleaved 16-byte pieces in its tile build (shipped) vs before (precolor): parity subset, A/B x2, LDS counters
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_backward.py tests/test_gpu_bench_config.py -x -q > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -3
bash tools/gpu_ab.sh precolor 2>&1 | tee $O/ab1.txt
bash tools/gpu_ab.sh precolor 2>&1 | tee $O/ab2.txt
python - <<'PY'
import json
for v in ("shipped","precolor"):
    d=json.load(open("gpurun_out/ab/bench_%s.json"%v)); print(v, "color_forward ms", d["kernels"]["color_forward"]["ms_per_step"])
PY
