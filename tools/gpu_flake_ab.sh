#!/bin/bash
# dev: bit-reproducibility of the inference forward (tests/flake_probe2.py) for the shipped library and variants: tools/gpu_flake_ab.sh N variant...
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
N=$1; shift
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in shipped "$@"; do
  [ $V != shipped ] && cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  echo "== $V"; (cd tests && timeout 600 python flake_probe2.py $N 2>&1 | tail -6)
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
