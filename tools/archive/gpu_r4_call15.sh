#!/bin/bash
# round 4, call 15: rocprofv3 passes again WITHOUT the supplementary two-plane steps in the profiled command (they doubled the weight-gradient bytes per "step")
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
timeout 900 bash tools/gpu_profile.sh r04 > gpurun_out/r04_profile.log 2>&1
timeout 600 bash tools/gpu_pmc_mfma.sh r04 > gpurun_out/r04_pmc_mfma.log 2>&1
tail -3 gpurun_out/r04_profile.log
