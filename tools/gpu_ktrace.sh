#!/bin/bash
# dev: rocprofv3 kernel trace of a short bench run; prints per-kernel totals and the individual launches of the tile kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-prof > $OUT/stats.log 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob("$OUT/stats/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tot = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    import re
    mm = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
    n = (mm.group(0) if mm else r["Kernel_Name"])[:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot[n][0] += d; tot[n][1] += 1
for n, (t, c) in sorted(tot.items(), key=lambda x: -x[1][0])[:22]:
    print("%-62s %9.3f ms %5d launches %8.3f ms each" % (n, t, c, t / c))
last = [r for r in rows if "k_agg_" in r["Kernel_Name"]][-12:]
for r in last:
    print(re.search(r"k_\w+", r["Kernel_Name"]).group(0), "grid", r.get("Grid_Size"), "vgpr", r.get("VGPR_Count"), "%.3f ms" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
