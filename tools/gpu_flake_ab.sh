#!/bin/bash
# dev: bit-reproducibility of the inference forward (tools/flake_probe2.py) for the shipped library and variants: tools/gpu_flake_ab.sh N variant...
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
N=$1; shift
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in "$@"; do
  [ $V != shipped ] && cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  echo "== $V"; (timeout 900 python tools/flake_probe2.py $N 2>&1 | tail -8)
  cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
done
