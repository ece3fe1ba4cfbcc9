#!/bin/bash
# rocprofv3 PMC pass over the bench: matrix-pipe busy and wave wait / issue fractions per kernel (counters only: no trace domains with --pmc)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-prof --no-variants"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc1 -o run -- $CMD > /tmp/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc2 -o run -- $CMD > /tmp/pmc2.log 2>&1
tail -2 /tmp/pmc1.log /tmp/pmc2.log
python - <<PY > $OUT/pmc_summary.txt
import csv, collections, glob
for d in ('/tmp/pmc1','/tmp/pmc2'):
    f = glob.glob(d+'/**/*counter_collection.csv', recursive=True)
    if not f: print('no csv in', d); continue
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in rows:
        k = r['Kernel_Name']
        for pat in ('k_agg_backward','k_agg_forward','k_wgrad_f16','k_wgrad_x0','k_color_forward','k_color_backward'):
            if pat in k:
                agg[pat][r['Counter_Name']] += float(r['Counter_Value']); n[pat][r['Counter_Name']] += 1
    for k,v in agg.items():
        print(k)
        for c,x in v.items(): print('    %-28s %.4e  per launch %.4e (%d)' % (c, x, x/max(n[k][c],1), n[k][c]))
        if 'GRBM_GUI_ACTIVE' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
            print('    MFMA busy / (GUI_ACTIVE / 8 XCD * 1024 SIMD): %.3f' % (v['SQ_VALU_MFMA_BUSY_CYCLES']/(v['GRBM_GUI_ACTIVE']/8*1024)))
        if 'SQ_WAVE_CYCLES' in v:
            w=v['SQ_WAVE_CYCLES']
            for c in ('SQ_WAIT_ANY','SQ_ACTIVE_INST_ANY','SQ_WAIT_INST_ANY'):
                if c in v: print('    %s / WAVE_CYCLES = %.3f' % (c, v[c]/w))
PY
cat $OUT/pmc_summary.txt
