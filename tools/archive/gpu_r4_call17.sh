#!/bin/bash
# round 4, call 17: the whole GPU suite (no -x)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c17; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
