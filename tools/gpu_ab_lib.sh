#!/bin/bash
# dev: bench the shipped library and a variant library (tools/_build/<name>.so) back to back on the same box
V=$1
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --cpu-rays 0 --steps 6 > gpurun_out/ab_shipped.json 2>/dev/null
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
timeout 300 python bench.py --cpu-rays 0 --steps 6 > gpurun_out/ab_variant.json 2>/dev/null
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
