#!/bin/bash
# dev: the query kernels' time for the shipped library and the variants given as arguments (tools/_build/<name>.so), configs in $CFGS
cd $GRAFT_REPO_ROOT
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in shipped "$@"; do
  [ $V != shipped ] && cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  for C in ${CFGS:-lego}; do echo -n "$V "; timeout 300 python tools/gpu_neighbors_ab.py $C 2>&1 | tail -1; done
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
