#!/bin/bash
# round 6, call b: K = 272 diagnosis, the new / changed tests under the default (backward-only e4m3), configs test under f16 cross terms as baseline
cd $GRAFT_REPO_ROOT
T=${1:-r6_b}; O=gpurun_out/$T; mkdir -p $O
python tools/gpu_mix_diag.py > $O/diag.log 2>&1; cat $O/diag.log | tail -40
timeout 900 python -m pytest tests/test_gpu_mix.py tests/test_gpu_bench_config.py -q -s -m gpu > $O/tests_new.log 2>&1; echo "new rc $?"; grep -E "^FAILED|passed|failed" $O/tests_new.log | tail
PNERF_MIX_MASK=0 timeout 900 python -m pytest tests/test_gpu_configs.py -q -s -m gpu > $O/configs_mask0.log 2>&1; echo "configs mask0 rc $?"; grep -E "^FAILED|passed|failed|points_|AssertionError" $O/configs_mask0.log | tail -20
timeout 900 python -m pytest tests/test_gpu_configs.py -q -s -m gpu > $O/configs_default.log 2>&1; echo "configs default rc $?"; grep -E "^FAILED|passed|failed|points_|AssertionError" $O/configs_default.log | tail -20
