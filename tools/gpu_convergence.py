"""dev / evidence: the convergence A/B of tests/test_gpu_zz_convergence.py with free parameters; prints one JSON line per run.
   python tools/gpu_convergence.py [steps = 2000] [runs per arithmetic = 2] [rays per step = 1024]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import convergence_case as C

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nruns = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rays = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda:0")
sc = C.scene()
for planes in (1, 2):
    for i in range(nruns):
        t = time.time()
        r = C.run(dev, steps, planes, rays_per_step=rays, sc=sc)
        torch.cuda.synchronize()
        r["seconds"] = round(time.time() - t, 1)
        print(json.dumps(r), flush=True)
