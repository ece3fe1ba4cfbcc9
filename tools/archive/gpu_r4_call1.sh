#!/bin/bash
# round 4, GPU call 1: packed-fp32 fault probes, new precision tests, bench with the fp32-class variant
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c1; mkdir -p $O
echo "== pkfma probe" | tee $O/pkfma.log
(timeout 300 tools/_build/pkfma_probe 100000 1; timeout 300 tools/_build/pkfma_probe 100000 0) 2>&1 | tee -a $O/pkfma.log
echo "== flake A/B"
timeout 1500 bash tools/gpu_flake_ab.sh 1500 vec vec_sched vec_nop shipped > $O/flake_ab.log 2>&1; tail -40 $O/flake_ab.log
echo "== convergence tool"
timeout 900 python tools/gpu_convergence.py 2000 2 > $O/convergence.jsonl 2> $O/convergence.err; cat $O/convergence.jsonl | cut -c1-600; tail -3 $O/convergence.err
echo "== tests"
timeout 1800 python -m pytest tests/test_gpu_convergence.py tests/test_gpu_bench_config.py tests/test_gpu_bench_dist.py tests/test_gpu_configs.py -x -q -s > $O/tests.log 2>&1; tail -60 $O/tests.log
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
