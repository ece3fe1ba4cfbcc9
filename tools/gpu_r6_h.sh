#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r6_h}; O=gpurun_out/$T; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu --deselect tests/test_gpu_zz_convergence.py::test_convergence_one_plane_vs_two_planes > $O/suite.log 2>&1; echo "suite rc $?"; tail -8 $O/suite.log
grep -n "^step " $O/suite.log | head -12
tools/gpu_r6_strong.sh $T
python tools/gpu_microbench.py > $O/microbench.json 2> $O/microbench.err; tail -2 $O/microbench.err; head -c 1500 $O/microbench.json
