#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c9; mkdir -p $O
timeout 1500 python tools/gpu_convergence.py 2000 12 > $O/convergence_12.jsonl 2> $O/convergence.err; tail -2 $O/convergence.err
python - <<'PY'
import json, numpy as np
rs=[json.loads(l) for l in open("gpurun_out/r4c9/convergence_12.jsonl")]
for key in ("train_mse","psnr_train","psnr_heldout","final_loss"):
    a=np.array([r[key] for r in rs if r["planes"]==1]); b=np.array([r[key] for r in rs if r["planes"]==2])
    se=np.sqrt(a.var(ddof=1)/a.size+b.var(ddof=1)/b.size)
    print(key, "one plane mean %.5g std %.3g | two planes mean %.5g std %.3g | diff %.3g = %.2f SE" % (a.mean(), a.std(ddof=1), b.mean(), b.std(ddof=1), a.mean()-b.mean(), (a.mean()-b.mean())/se))
print("seconds per run", np.mean([r["seconds"] for r in rs]))
PY
timeout 1800 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_configs.py tests/test_gpu_pkfma_probe.py -q -s > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -40
