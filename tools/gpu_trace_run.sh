#!/bin/bash
# dev: per-phase timeline of the two tile kernels with the PN_PHASE_TRACE build (tools/_build/trace.so)
cd $GRAFT_REPO_ROOT
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
cp tools/_build/trace.so pointnerf_amd/libpnerf_hip.so
timeout 600 python tools/gpu_phase_trace.py > gpurun_out/phase_trace.json 2> gpurun_out/phase_trace.err
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
tail -3 gpurun_out/phase_trace.err; cat gpurun_out/phase_trace.json
