#!/bin/bash
# round 4, call 14: the round's evidence run on the shipped build (bench lines, rocprofv3 stats, HBM / MFMA counters, microbench) + phase trace
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
bash tools/gpu_round_profile.sh r4
bash tools/gpu_trace_run.sh > gpurun_out/r4/phase_trace.log 2>&1; cp gpurun_out/phase_trace.json gpurun_out/r4/ 2>/dev/null
timeout 200 bash tools/gpu_r4_call10.sh > gpurun_out/r4/extract2d.log 2>&1; cp gpurun_out/r4c10/embed_time.json gpurun_out/r4/ 2>/dev/null; tail -3 gpurun_out/r4/extract2d.log
