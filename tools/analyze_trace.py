"""dev: per-step GPU timeline from a rocprofv3 kernel_trace.csv: busy time by kernel family and idle gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by k_probe launches
starts = [i for i, r in enumerate(rows) if "k_probe" in r["Kernel_Name"]]
if len(starts) < 3:
    sys.exit("need >=3 steps")
a, b = starts[-2], starts[-1]          # last full step
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = collections.defaultdict(float); cnt = collections.Counter()
last_end = t0; idle = 0.0; gaps = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    fam = "pnerf:" + n.split("::")[-1].split("(")[0].split("<")[0] if "anonymous namespace)::k_" in n and "at::" not in n else ("torch:" + n.split("<")[0].split("::")[-1][:40])
    busy[fam] += (e - s) / 1e6; cnt[fam] += 1
    if s > last_end:
        idle += (s - last_end) / 1e6
        gaps.append(((s - last_end) / 1e6, fam))
    last_end = max(last_end, e)
print("step wall %.2f ms, idle %.2f ms, kernels %d" % ((t1 - t0) / 1e6, idle, len(seg)))
for k, v in sorted(busy.items(), key=lambda x: -x[1])[:22]:
    print("  %-58s %8.3f ms  x%d" % (k, v, cnt[k]))
print("largest gaps (ms, before kernel):", sorted(gaps, reverse=True)[:8])
