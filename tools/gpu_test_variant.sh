#!/bin/bash
# dev: run a subset of the GPU parity tests against variant libraries (tools/_build/<name>.so)
cd $GRAFT_REPO_ROOT
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in "$@"; do
  cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  echo "== $V"; timeout 300 python -m pytest tests/test_gpu_backward.py tests/test_gpu_render.py tests/test_gpu_train_steps.py -q -x 2>&1 | tail -2
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
