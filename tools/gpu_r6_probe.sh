#!/bin/bash
# round 6: tools/mx_probe on the device (built by hipcc here, the binary travels): gpurun_out/<tag>/mx_probe.jsonl
cd $GRAFT_REPO_ROOT
T=$1; O=gpurun_out/$T; mkdir -p $O
timeout 300 tools/_build/mx_probe $2 > $O/mx_probe.jsonl 2> $O/mx_probe.err; echo "rc $?"
grep rate $O/mx_probe.jsonl
