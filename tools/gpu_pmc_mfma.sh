#!/bin/bash
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-prof"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc1 -o run -- $CMD > /tmp/pmc1.log 2>&1
tail -3 /tmp/pmc1.log
python - <<'PY'
import csv, collections, glob
f = glob.glob('/tmp/pmc1/**/*counter_collection.csv', recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r['Kernel_Name']
    for pat in ('k_agg_backward','k_agg_forward','k_wgrad_lds<4, 2','k_color_forward','k_color_backward'):
        if pat in k:
            agg[pat][r['Counter_Name']] += float(r['Counter_Value']); 
            if r['Counter_Name']=='GRBM_GUI_ACTIVE': n[pat]+=1
for k,v in agg.items():
    print(k, 'launches', n[k])
    for c,x in v.items(): print('    %-28s %.4e  per launch %.4e' % (c, x, x/max(n[k],1)))
    if 'GRBM_GUI_ACTIVE' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        print('    MFMA busy / (GUI_ACTIVE * 256 CU * 4 SIMD): %.3f' % (v['SQ_VALU_MFMA_BUSY_CYCLES']/(v['GRBM_GUI_ACTIVE']*1024)))
PY
