"""evidence (round 6): the convergence A/B of tests/convergence_case.py (teacher / student, 2 000 steps, held-out PSNR) for the three cross-term
arithmetics of the aggregator's tile GEMMs: f16 everywhere (rounds 2-5), e4m3 in the backward's input-gradient chain (shipped), e4m3 in the
training forward too (optional).  Weight gradients: the shipped one-plane form in all three.
   python tools/gpu_convergence_ct.py [steps = 2000] [runs per arithmetic = 6]  -> one JSON line per run + a summary line"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import convergence_case as C
from pointnerf_amd import ops

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nruns = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
sc = C.scene()
arms = {"f16 cross terms everywhere": (16, 4), "e4m3 in the input-gradient chain (shipped)": (8, 4), "e4m3 in training forward and backward": (8, 7)}
res = {k: [] for k in arms}
for name, (bits, where) in arms.items():
    ops.set_cross_terms(bits, where=where)
    for i in range(nruns):
        t = time.time()
        r = C.run(dev, steps, 1, sc=sc)
        torch.cuda.synchronize()
        r["seconds"] = round(time.time() - t, 1); r["arithmetic"] = name
        r.pop("loss_curve", None)
        res[name].append(r)
        print(json.dumps(r), flush=True)
ops.set_cross_terms(8, where=4)
summary = {"steps": steps, "runs_per_arithmetic": nruns}
base = res["f16 cross terms everywhere"]
for name, rs in res.items():
    out = {}
    for key in ("psnr_heldout", "psnr_train", "train_mse"):
        a, b = np.array([r[key] for r in rs]), np.array([r[key] for r in base])
        se = float(np.sqrt(a.var(ddof=1) / a.size + b.var(ddof=1) / b.size))
        out[key] = {"mean": float(a.mean()), "std": float(a.std(ddof=1)), "difference_to_f16": float(a.mean() - b.mean()), "standard_error_of_the_difference": se}
    out["psnr_heldout_before"] = rs[0]["psnr_heldout_before"]
    summary[name] = out
print(json.dumps({"summary": summary}))
