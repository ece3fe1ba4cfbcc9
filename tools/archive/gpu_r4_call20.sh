#!/bin/bash
# round 4, call 20: the round's evidence run again on the final build (fused colour loss, step enqueued before its host read)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
bash tools/gpu_round_profile.sh r04
bash tools/gpu_trace_run.sh > gpurun_out/r04/phase_trace.log 2>&1; cp gpurun_out/phase_trace.json gpurun_out/r04/ 2>/dev/null
