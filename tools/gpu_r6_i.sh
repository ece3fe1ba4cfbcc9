#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r6_i}; O=gpurun_out/$T; mkdir -p $O
PNERF_MIX_MASK=5 timeout 1700 python -m pytest tests -q -m gpu --deselect tests/test_gpu_zz_convergence.py::test_convergence_one_plane_vs_two_planes > $O/suite_mask5.log 2>&1; echo "suite mask5 rc $?"; grep -E "^FAILED|passed|failed" $O/suite_mask5.log | tail -30
PNERF_MIX_MASK=5 python bench.py --render-only --steps 10 --warmup 3 --cpu-rays 0 > $O/render_only_mask5.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/render_only_mask5.json')); print('render-only mask5', d['value'], d['ms_per_step'])"
