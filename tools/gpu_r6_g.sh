#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r6_g}; O=gpurun_out/$T; mkdir -p $O
python bench.py --steps 12 --warmup 4 --no-variants > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<P
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], {n:round(v["ms_per_step"],2) for n,v in d["kernels"].items() if v.get("ms_per_step",0)>0.25})
P
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_zz_convergence.py::test_convergence_one_plane_vs_two_planes > $O/suite.log 2>&1; echo "suite rc $?"; tail -5 $O/suite.log
grep -n "^step " $O/suite.log | head -12
