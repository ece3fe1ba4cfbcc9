#!/bin/bash
# round 5: the whole device suite in the driver's form (-x, collection order of tests/conftest.py), per-test durations, then smoke and the bench line.
# Usage: bash tools/gpu_r5_suite.sh <tag> [pytest args]   -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
T=$1; shift; O=gpurun_out/$T; mkdir -p $O
( hostname; rocm-smi --showproductname 2>/dev/null | grep -i "card series" | head -1; nproc ) > $O/box.txt 2>&1
S=$(date +%s)
timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 -rs "$@" > $O/tests.log 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - S )) s" | tee -a $O/tests.log
grep -v amdgpu.ids $O/tests.log | tail -45
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 600 python bench.py > $O/bench_lego.json 2>$O/bench_lego.err; tail -c 600 $O/bench_lego.json
