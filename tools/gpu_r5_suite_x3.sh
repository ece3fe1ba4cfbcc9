#!/bin/bash
# round 5: the device suite three times back to back in the driver's form on one lease (flake hunting): gpurun_out/<tag>/run{1,2,3}.log
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/$1; mkdir -p $O
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/run$i.log 2>&1
  echo "run $i rc=$?: $(grep -v amdgpu.ids $O/run$i.log | tail -1)"
done
