#!/bin/bash
# round 5: clock and power sampled (rocm-smi, 0.3 s) while tools/_build/mfma_power_probe runs its operand modes one after the other (~2 s each):
# the clock the power management grants the matrix pipe by operand activity.  -> gpurun_out/<tag>/mfma_clock_watch.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 300 tools/_build/mfma_power_probe > $O/probe.jsonl 2>&1 ) &
BP=$!
while kill -0 $BP 2>/dev/null; do
  L=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' ')
  echo "$L"
  sleep 0.3
done > $O/mfma_clock_watch.txt
wait $BP
cat $O/probe.jsonl; cat $O/mfma_clock_watch.txt
