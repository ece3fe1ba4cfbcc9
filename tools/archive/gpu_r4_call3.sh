#!/bin/bash
# round 4, GPU call 3: probes with VALU / LDS / mixed neighbours, flake A/B of the in-branch guard variants
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c3; mkdir -p $O
echo "== pkfma probes" | tee $O/pkfma.log
for bg in 2 3 4; do timeout 300 tools/_build/pkfma_probe 60000 $bg; timeout 300 tools/_build/pkfma_probe replay 60000 $bg; done 2>&1 | tee -a $O/pkfma.log
echo "== flake A/B"
timeout 2400 bash tools/gpu_flake_ab.sh 1500 vec_g3 vec_g4 vec_nobr > $O/flake_ab.log 2>&1; grep -v amdgpu.ids $O/flake_ab.log | tail -40
