#!/bin/bash
# round 4, call 24: weight-fragment prefetch distance of the tile GEMMs (PN_WPF 1 / 2 (shipped) / 3) now that the first chunks are requested in front of the barrier
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c24; mkdir -p $O
bash tools/gpu_ab.sh wpf1 wpf3 2>&1 | tee $O/ab1.txt
bash tools/gpu_ab.sh wpf1 wpf3 2>&1 | tee $O/ab2.txt
