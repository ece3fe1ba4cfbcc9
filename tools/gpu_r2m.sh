#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_query.py -q -p no:cacheprovider 2>&1 | tail -30 > $O/tests.log); tail -3 $O/tests.log
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_lego.json 2>$O/bench_lego.err
for C in scannet barn; do
  timeout 600 python bench.py --config $C --steps 4 --warmup 2 > $O/bench_$C.json 2>$O/bench_$C.err
done
for f in lego scannet barn; do python - <<PY
import json
try:
    d=json.load(open("$O/bench_$f.json")); k=d["kernels"]
    print("%-8s %.0f rays/s %.2f ms (median %.2f) neighbors %.3f probe %.3f grid %s" % ("$f", d["value"], d["ms_per_step"], d["median_ms_per_step"], k["neighbors"]["ms_per_step"], k["probe"]["ms_per_step"], k.get("grid")))
except Exception as e: print("$f", "ERR", e)
PY
done
