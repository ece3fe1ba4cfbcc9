#!/bin/bash
# the round's evidence run: bench JSON lines of the four configurations (+ render only), rocprofv3 kernel stats, HBM counters (their own
# passes), matrix-pipe / wait counters, the micro-benchmarks.  Usage: bash tools/gpu_round_profile.sh <tag>  -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
T=$1; O=gpurun_out/$T; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_lego.json 2>$O/bench_lego.err
timeout 300 python bench.py --steps 10 --warmup 2 --render-only --cpu-rays 0 > $O/bench_render_only.json 2>$O/bench_render_only.err
timeout 300 python bench.py --steps 10 --warmup 2 --render-only --inference-products 2 --cpu-rays 0 > $O/bench_render_only_2products.json 2>/dev/null
for C in chair scannet barn; do
  timeout 600 python bench.py --config $C --steps 5 --warmup 2 --cpu-rays 0 > $O/bench_$C.json 2>$O/bench_$C.err
done
timeout 900 bash tools/gpu_profile.sh $T > $O/profile.log 2>&1
timeout 600 bash tools/gpu_pmc_mfma.sh $T > $O/pmc_mfma.log 2>&1
timeout 600 python tools/gpu_microbench.py > $O/microbench.json 2>$O/microbench.err
for f in lego render_only chair scannet barn; do python - <<PY
import json
try:
    d=json.load(open("$O/bench_$f.json")); k=d.get("kernels",{})
    print("%-12s %.0f rays/s %.2f ms (median %.2f)" % ("$f", d["value"], d["ms_per_step"], d.get("median_ms_per_step", 0)), {n: round(v["ms_per_step"],2) for n,v in k.items() if v["ms_per_step"] > 0.3})
except Exception as e: print("$f", "ERR", e)
PY
done
tail -5 $O/microbench.err
