#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r6_d}; O=gpurun_out/$T; mkdir -p $O
python tools/gpu_mix_diag.py > $O/diag.log 2>&1; tail -14 $O/diag.log
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
cp tools/_build/trace.so pointnerf_amd/libpnerf_hip.so
timeout 600 python tools/gpu_phase_trace.py > $O/phase_trace.json 2> $O/phase_trace.err
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
tail -3 $O/phase_trace.err; python - <<P
import json
d=json.load(open("$O/phase_trace.json"))
for k in ("forward","backward"):
    print(k, d[k].get("tile_iteration_us_mean"), d[k].get("gemm_us_per_tile"))
    for n,v in d[k]["phase_us_mean_p90"].items(): print("   %-70s %6.2f %6.2f" % (n, v[0], v[1]))
P
