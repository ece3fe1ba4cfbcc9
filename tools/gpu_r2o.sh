#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2o; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider 2>&1 | tail -40 > $O/tests.log); tail -3 $O/tests.log
timeout 400 python bench.py --steps 20 --warmup 3 > $O/bench_lego.json 2>$O/bench_lego.err
python - <<PY
import json
d=json.load(open("$O/bench_lego.json")); k=d["kernels"]
print("%.0f rays/s %.2f ms (median %.2f)" % (d["value"], d["ms_per_step"], d["median_ms_per_step"]), {n: round(v["ms_per_step"],3) for n,v in k.items()})
PY
