#!/bin/bash
# instruction-cache counters of the tile kernels (their loop bodies are 100-200 KB of straight-line code)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQC_|INST_CACHE|WAIT_IFETCH" | head -60 > $OUT/icache_counters.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-prof"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmci -o run -- $CMD > /tmp/pmci.log 2>&1
tail -3 /tmp/pmci.log
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d /tmp/pmcj -o run -- $CMD > /tmp/pmcj.log 2>&1
tail -3 /tmp/pmcj.log
python - <<PY > $OUT/icache_summary.txt
import csv, collections, glob
for d in ('/tmp/pmci','/tmp/pmcj'):
    f = glob.glob(d+'/**/*counter_collection.csv', recursive=True)
    if not f: print('no csv in', d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        for pat in ('k_agg_backward','k_agg_forward','k_wgrad_f16','k_color_forward','k_color_backward'):
            if pat in k: agg[pat][r['Counter_Name']] += float(r['Counter_Value'])
    for k,v in agg.items():
        print(k, {c: '%.4e' % x for c,x in v.items()})
PY
cat $OUT/icache_counters.txt | head -40; cat $OUT/icache_summary.txt
