#!/bin/bash
# dev: what distinguishes the pool's slow boxes?  Power cap / clocks / partition mode next to a short bench.
cd $GRAFT_REPO_ROOT
{
/opt/rocm/bin/rocm-smi --showpower --showmaxpower --showclocks --showmemuse --showperflevel --showcomputepartition --showmemorypartition 2>&1 | grep -v "^=\|^$" | head -40
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Compute Unit|Max Clock|Marketing" | head -6
timeout 200 python bench.py --cpu-rays 0 --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BENCH', round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items() if v['ms_per_step']>20})"
/opt/rocm/bin/rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | head -8
} > gpurun_out/box_info_$(date +%s).txt 2>&1
tail -n 60 gpurun_out/box_info_*.txt | tail -60
