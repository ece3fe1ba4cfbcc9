#!/bin/bash
# dev: kernel-trace the shipped library and variants back to back: tools/gpu_ktrace_ab.sh tag variant...
cd $GRAFT_REPO_ROOT
T=$1; shift
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in shipped "$@"; do
  [ $V != shipped ] && cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  echo "== $V"; bash tools/gpu_ktrace.sh ${T}_$V 2>&1 | grep -E "k_agg|k_wgrad|k_color" | head -16
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
