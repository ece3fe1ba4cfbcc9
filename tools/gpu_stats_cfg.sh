#!/bin/bash
# dev: rocprofv3 kernel stats of one bench configuration: bash tools/gpu_stats_cfg.sh <config> <tag>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$2; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python $GRAFT_REPO_ROOT/bench.py --config $1 --steps 4 --warmup 1 --cpu-rays 0 --no-prof > $OUT/stats.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/stats/run_kernel_stats.csv")))
for r in rows[:28]: print("%8.3f ms total %5s calls  %s" % (float(r["TotalDurationNs"])/1e6, r["Calls"], r["Name"][:100]))
PY
tail -2 $OUT/stats.log
