#!/bin/bash
# rocprofv3 passes of the bench command: kernel trace + stats, then HBM counters in their own runs.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-prof --no-variants"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -type f | head -30
du -sh $OUT
