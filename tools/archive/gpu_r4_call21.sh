#!/bin/bash
# round 4, call 21: contiguous runs of tiles per workgroup (-DPN_TILE_BLOCKED) against the strided assignment: parity of the variant, A/B x2
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c21; mkdir -p $O
cp pointnerf_amd/libpnerf_hip.so /tmp/ship.so; cp tools/_build/blocked.so pointnerf_amd/libpnerf_hip.so
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_backward.py tests/test_gpu_bench_config.py -x -q > $O/tests_blocked.log 2>&1; grep -v amdgpu.ids $O/tests_blocked.log | tail -3
cp /tmp/ship.so pointnerf_amd/libpnerf_hip.so
bash tools/gpu_ab.sh blocked 2>&1 | tee $O/ab1.txt
bash tools/gpu_ab.sh blocked 2>&1 | tee $O/ab2.txt
