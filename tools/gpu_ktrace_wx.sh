#!/bin/bash
cd $GRAFT_REPO_ROOT
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in shipped "$@"; do
  [ $V != shipped ] && cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  echo "== $V"; PNERF_BENCH_ALLOW_NAN=1 bash tools/gpu_ktrace.sh wx_$V 2>&1 | grep -E "k_wgrad" | head -6
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
