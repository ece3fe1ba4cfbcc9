#!/bin/bash
# round 5: tools/role_probe on the device (built by hipcc here, the binary travels): gpurun_out/<tag>/role_probe.jsonl
cd $GRAFT_REPO_ROOT
T=$1; O=gpurun_out/$T; mkdir -p $O
timeout 600 tools/_build/role_probe | tee $O/role_probe.jsonl
