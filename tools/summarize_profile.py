"""Turn gpurun_out/<tag>/ (rocprofv3 csv output of tools/gpu_profile.sh) into the small tracked files under profiles/."""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1]
src, dst = os.path.join("gpurun_out", tag), "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats", "run_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
short = {"k_agg_backward": "agg_backward", "k_agg_forward": "agg_forward", "k_color_forward": "color_forward",
         "k_color_backward": "color_backward", "k_wgrad_lds": "wgrad", "k_wgrad_f16": "wgrad", "k_wgrad_x0": "wgrad", "k_wgrad<": "wgrad", "k_neighbors": "neighbors", "k_probe": "probe"}
def agg(path):
    out = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        for pat, name in short.items():
            if pat in r["Kernel_Name"]:
                out[name][0] += float(r["Counter_Value"]); out[name][1] += 1
    return out
f, w = agg(os.path.join(src, "pmc_fetch", "run_counter_collection.csv")), agg(os.path.join(src, "pmc_write", "run_counter_collection.csv"))
summary, traffic = {}, {}
steps = max(f.get("color_backward", [0, 0])[1], 1)   # one colour-backward launch per step the command ran (timed, warm-up and supplementary steps)
for k in sorted(set(f) | set(w)):
    fk, wk = f.get(k, [0, 1]), w.get(k, [0, 1])
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128 B request of a wide coalesced
    # read, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported (uncalibrated).
    per_step_bytes = (2.0 * fk[0] + wk[0]) * 1024.0 / steps
    launches_per_step = max(fk[1], wk[1]) / steps
    if k in ("neighbors", "probe"):        # one launch per step; bench.py also runs them outside its steps (arena sizing): average per launch
        per_step_bytes = (2.0 * fk[0] / max(fk[1], 1) + wk[0] / max(wk[1], 1)) * 1024.0
        launches_per_step = 1.0
    summary[k] = {"FETCH_SIZE_KiB_total": fk[0], "WRITE_SIZE_KiB_total": wk[0], "launches": max(fk[1], wk[1]),
                  "hbm_bytes_per_step_corrected": per_step_bytes, "launches_per_step": launches_per_step}
    traffic[k] = per_step_bytes
import subprocess
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
    dirty = bool(subprocess.check_output(["git", "status", "--porcelain", "--", "pointnerf_amd/csrc", "bench.py"]).decode().strip())
except Exception:
    commit, dirty = None, None
# (the command of tools/gpu_profile.sh's --pmc passes; the library is the one built from `commit`)
traffic["_source"] = {"commit": commit, "uncommitted_changes_in_the_library_or_bench": dirty, "tag": tag,
                      "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-prof --no-variants",
                      "correction": "(2 x FETCH_SIZE + WRITE_SIZE) KiB per step (MI355X_MICROARCH.md, HBM section)"}
json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
