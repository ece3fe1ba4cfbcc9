#!/bin/bash
# round 4, call 19: epilogue arithmetic in front of the barrier (shipped) vs the build before (preepi): parity subset, A/B x2
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_backward.py tests/test_gpu_bench_config.py tests/test_gpu_reproducible.py -x -q > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -3
bash tools/gpu_ab.sh preepi 2>&1 | tee $O/ab1.txt
bash tools/gpu_ab.sh preepi 2>&1 | tee $O/ab2.txt
