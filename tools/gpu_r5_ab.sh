#!/bin/bash
# round 5 A/B on one box: parity tests of the shipped build first (a variant that computes garbage clocks higher), then tools/gpu_ab.sh <variants>, twice
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
T=$1; shift; O=gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bench_config.py tests/test_gpu_train_steps.py tests/test_gpu_level1.py tests/test_gpu_reproducible.py -x -q > $O/tests.log 2>&1
grep -v amdgpu.ids $O/tests.log | tail -4
bash tools/gpu_ab.sh "$@" 2>&1 | tee $O/ab1.txt
bash tools/gpu_ab.sh "$@" 2>&1 | tee $O/ab2.txt
