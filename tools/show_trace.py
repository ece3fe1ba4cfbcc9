"""dev: print a gpu_phase_trace.py JSON as a table"""
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("forward", "backward"):
    r = d[k]
    ng = sum(v[0] for n, v in r["phase_us_mean_p90"].items() if "GEMM" not in n and "G(" not in n)
    print(k, "iter", r["tile_iteration_us_mean"], "gemm", r["gemm_us_per_tile"], "non-gemm %.1f" % ng, r.get("time_frac_in_gemm_phase"))
    for n, v in r["phase_us_mean_p90"].items():
        print("   %-34s %6.2f %6.2f" % (n, v[0], v[1]))
