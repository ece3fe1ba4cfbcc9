#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2n; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider 2>&1 | tail -40 > $O/tests.log); tail -3 $O/tests.log
timeout 400 python bench.py --steps 20 --warmup 3 > $O/bench_lego.json 2>$O/bench_lego.err
for C in chair scannet barn; do
  timeout 600 python bench.py --config $C --steps 5 --warmup 2 > $O/bench_$C.json 2>$O/bench_$C.err
done
for f in lego chair scannet barn; do python - <<PY
import json
try:
    d=json.load(open("$O/bench_$f.json")); k=d["kernels"]; c=d["config"]
    print("%-8s %.0f rays/s %.2f ms (median %.2f) rows/s %.3g fwd %.2f bwd %.2f wgrad %.2f saved %.1f GB chunks %s roofline %s %.3f" % ("$f", d["value"], d["ms_per_step"], d["median_ms_per_step"], d["neighbor_rows_per_s"], k["agg_forward"]["ms_per_step"], k.get("agg_backward",{}).get("ms_per_step",0), k.get("wgrad",{}).get("ms_per_step",0), c["saved_activation_bytes_per_step"]/2**30, c["backward_ray_chunks"], d["roofline"]["kernel"], d["roofline"]["frac"]))
except Exception as e: print("$f", "ERR", e)
PY
done
tail -3 $O/bench_barn.err
