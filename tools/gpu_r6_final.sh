#!/bin/bash
# the round's closing run at HEAD: build check, smoke(), the device suite in the driver's form, then the evidence passes
cd $GRAFT_REPO_ROOT
T=${1:-r06}; O=gpurun_out/$T; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; echo "suite rc $?"; tail -4 $O/suite.log
bash tools/gpu_round_profile.sh $T > $O/round_profile.log 2>&1; tail -8 $O/round_profile.log
