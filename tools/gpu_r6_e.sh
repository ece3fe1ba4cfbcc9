#!/bin/bash
# after the scale-block fix: tile GEMM tests, forward modes, and the parity tests with e4m3 everywhere (mask 7) again
cd $GRAFT_REPO_ROOT
T=${1:-r6_e}; O=gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mix.py -q -s -m gpu > $O/test_mix.log 2>&1; echo "mix rc $?"; grep -E "^FAILED|passed|failed| train=" $O/test_mix.log | tail -30
PNERF_MIX_MASK=7 timeout 1500 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_render.py tests/test_gpu_backward.py tests/test_gpu_trig.py tests/test_gpu_train_steps.py tests/test_gpu_configs.py -q -s -m gpu > $O/tests_mask7.log 2>&1
echo "mask 7 rc $?"; grep -E "^FAILED|passed|failed" $O/tests_mask7.log | tail -30
grep -E "forward max abs|max abs errors|points beyond" $O/tests_mask7.log | cut -c1-250 | head -30
