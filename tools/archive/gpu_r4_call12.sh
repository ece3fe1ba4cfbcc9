#!/bin/bash
# round 4, call 12: PMC counters of the two tile kernels in the 8-wave organisation (shipped build of this call) and in the 4-wave one (orgA)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
bash tools/gpu_pmc_mfma.sh r4c12/orgB > /dev/null 2>&1
cp pointnerf_amd/libpnerf_hip.so /tmp/s.so; cp tools/_build/orgA.so pointnerf_amd/libpnerf_hip.so
bash tools/gpu_pmc_mfma.sh r4c12/orgA > /dev/null 2>&1
cp /tmp/s.so pointnerf_amd/libpnerf_hip.so
for v in orgB orgA; do echo "=== $v"; grep -A22 "k_agg_forward\|k_agg_backward" gpurun_out/r4c12/$v/pmc_summary.txt | grep -v "k_wgrad\|k_color" | head -60; done
