#!/bin/bash
# round 4, GPU call 2: verbatim replay of the faulty tail, flake A/B of the vectorised variants, precision tests
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c2; mkdir -p $O
echo "== pkfma replay" | tee $O/pkfma_replay.log
(timeout 300 tools/_build/pkfma_probe replay 100000 1; timeout 300 tools/_build/pkfma_probe replay 100000 0) 2>&1 | tee -a $O/pkfma_replay.log
echo "== flake A/B"
timeout 1500 bash tools/gpu_flake_ab.sh 1500 vec vec_sched vec_nop > $O/flake_ab.log 2>&1; grep -v amdgpu.ids $O/flake_ab.log | tail -40
echo "== tests"
timeout 2400 python -m pytest tests/test_gpu_convergence.py tests/test_gpu_bench_config.py tests/test_gpu_bench_dist.py tests/test_gpu_configs.py -q -s > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -120
cp gpurun_out/convergence_ab.json $O/ 2>/dev/null
