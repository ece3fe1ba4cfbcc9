#!/bin/bash
# dev: PMC counters of the split-bf16 weight-gradient kernel (two passes, counters only)
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-prof"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmcw1 -o run -- $CMD > /tmp/pmcw1.log 2>&1
tail -2 /tmp/pmcw1.log
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcw2 -o run -- $CMD > /tmp/pmcw2.log 2>&1
tail -2 /tmp/pmcw2.log
python - <<'PY'
import csv, collections, glob
for d in ('/tmp/pmcw1', '/tmp/pmcw2'):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        print(d, 'no csv'); continue
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in rows:
        k = r['Kernel_Name']
        for pat in ('k_wgrad_b3<256', 'k_agg_backward<true'):
            if pat in k:
                agg[pat][r['Counter_Name']] += float(r['Counter_Value'])
                if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[pat] += 1
    for k, v in agg.items():
        print(k, 'launches', n[k])
        for c, x in v.items(): print('    %-28s per launch %.4e' % (c, x / max(n[k], 1)))
PY
