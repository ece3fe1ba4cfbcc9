#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r6_c}; O=gpurun_out/$T; mkdir -p $O
python tools/gpu_mix_diag.py > $O/diag.log 2>&1; tail -12 $O/diag.log
timeout 900 python -m pytest tests/test_gpu_configs.py -q -s -m gpu > $O/configs_default.log 2>&1; echo "configs default rc $?"; grep -E "^FAILED|passed|failed" $O/configs_default.log | tail -5
python bench.py --steps 10 --warmup 4 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<P
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], {n:round(v["ms_per_step"],2) for n,v in d["kernels"].items() if v.get("ms_per_step",0)>0.3})
for k,v in d["config"].items():
    if k.endswith("_variant"): print(k, round(v["ms_per_step"],2), round(v["value"]))
print(d["roofline"]["frac"], d["roofline"]["f16_products_per_multiply_add"], d["roofline"]["kernel"])
P
