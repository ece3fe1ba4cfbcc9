#!/bin/bash
# Runs tests/test_gpu_overlay.py on the GPU box: the reference checkout is not part of this repository and the box has none, so its Python
# (4 MB, no data) is shipped for the duration of ONE gpurun call in .refship/ (git-ignored, never committed) and removed afterwards.
#   tools/gpu_overlay_run.sh            -> gpurun_out/overlay_gpu.log ; copy it to profiles/rNN_overlay_gpu.log
set -e
cd "$(dirname "$0")/.."
REF=${POINTNERF_REFERENCE_SRC:-/root/reference}
rm -rf .refship && mkdir .refship
tar -C "$REF" --exclude=.git --exclude=images -cf - . | tar -C .refship -xf -
trap 'rm -rf .refship' EXIT
/usr/local/graft/bin/gpurun --timeout ${TIMEOUT:-900} -- 'mkdir -p gpurun_out; export POINTNERF_REFERENCE=$GRAFT_REPO_ROOT/.refship; '"${EXTRA_CMD:-true}"'; python -m pytest tests/test_gpu_overlay.py -x -q -s > gpurun_out/overlay_gpu.log 2>&1; echo overlay rc=$? >> gpurun_out/overlay_gpu.log; tail -5 gpurun_out/overlay_gpu.log'
