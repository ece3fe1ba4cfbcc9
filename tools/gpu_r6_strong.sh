#!/bin/bash
# round 6: what one rank of an 8-/4-/2-GPU strong-scaling run (configs[2]: 65 536 rays split) does per step, priced on ONE GPU with every collective
# executed on a one-rank RCCL group (--force-collectives): 8 192 / 16 384 / 32 768 / 65 536 rays per rank, dense / sparse / zero1 exchange
cd $GRAFT_REPO_ROOT
T=${1:-r6_strong}; O=gpurun_out/$T; mkdir -p $O
for R in 8192 16384 32768 65536; do
  for X in dense sparse zero1; do
    if [ $X = zero1 ]; then F="--zero1"; else F="--point-grads $X"; fi
    python bench.py --steps 12 --warmup 4 --rays $R --force-collectives --no-variants --cpu-rays 0 $F > $O/b_${R}_$X.json 2> $O/b_${R}_$X.err || tail -3 $O/b_${R}_$X.err
  done
done
python - <<P
import json, glob
out = {"what": "one rank's step of BASELINE.json configs[2] (the 65 536-ray batch split over N GPUs) on ONE MI355X with every collective of the step executed on a one-rank RCCL group (bench.py --force-collectives): kernels / Adam / everything else, per rays-per-rank and point-gradient exchange", "rows": []}
for R in (8192, 16384, 32768, 65536):
    for X in ("dense", "sparse", "zero1"):
        try:
            d = json.load(open("$O/b_%d_%s.json" % (R, X)))
        except Exception as e:
            out["rows"].append({"rays_per_rank": R, "exchange": X, "error": repr(e)}); continue
        k = d["kernels"]
        lib = sum(v["ms_per_step"] for v in k.values())
        row = {"rays_per_rank": R, "ranks_of_65536": 65536 // R, "exchange": d["config"]["point_grad_exchange"], "ms_per_step": d["ms_per_step"],
               "ms_library_kernels": lib, "ms_adam": k.get("adam", {}).get("ms_per_step"), "ms_outside_library_kernels": d["ms_outside_library_kernels"],
               "ms_exchange_on_main_stream_one_rank": d["config"]["ms_allreduce_exposed_by_rank"], "rays_per_s_this_rank": d["value"],
               "kernels_ms": {n: round(v["ms_per_step"], 3) for n, v in k.items() if v["ms_per_step"] > 0.05}}
        out["rows"].append(row)
        print(R, X, round(d["ms_per_step"], 2), "lib", round(lib, 2), "adam", row["ms_adam"], "outside", round(row["ms_outside_library_kernels"], 2), "exposed", row["ms_exchange_on_main_stream_one_rank"])
json.dump(out, open("$O/strong_scaling_rank_cost.json", "w"), indent=1)
P
