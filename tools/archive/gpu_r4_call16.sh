#!/bin/bash
# round 4, call 16: the whole GPU suite on the shipped build; then the step enqueued before its host read (default) against PNERF_SPECULATE=0,
# and kernel-trace gaps of one step
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c16; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -6
for i in 1 2; do
PNERF_SPECULATE=0 timeout 300 python bench.py --cpu-rays 0 --steps 20 --no-fp32-class-variant > $O/bench_wait_$i.json 2>/dev/null
timeout 300 python bench.py --cpu-rays 0 --steps 20 --no-fp32-class-variant > $O/bench_spec_$i.json 2>/dev/null
done
timeout 300 python bench.py --config chair --cpu-rays 0 --steps 20 > $O/bench_chair.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4c16/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"]), "rays/s", round(d["ms_per_step"],3), "ms  outside library kernels", round(d.get("ms_outside_library_kernels",0),3))
    except Exception as e: print(f, "ERR", e)
PY
