#!/bin/bash
# dev: bench the shipped library and variant libraries (tools/_build/<name>.so) back to back on the same box
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --cpu-rays 0 --steps 6 > gpurun_out/ab_shipped.json 2>/dev/null
timeout 300 python bench.py --cpu-rays 0 --steps 6 > gpurun_out/ab_shipped2.json 2>/dev/null
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in "$@"; do
  cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  timeout 300 python bench.py --cpu-rays 0 --steps 6 > gpurun_out/ab_$V.json 2>/dev/null
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
