#!/bin/bash
# round 5: the fused two-plane weight-gradient kernel: parity of the fp32-class mode (bench-config gradients against float64, 200 steps inside the
# oracle ensemble, the bench's own variant line), then the fp32-class variant timed with the shipped build and with a variant library
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/$1; shift; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_backward.py "tests/test_gpu_zz_convergence.py::test_200_steps_inside_the_oracle_ensemble" "tests/test_gpu_bench_dist.py::test_bench_json_contract_single_gpu" -x -q -s > $O/tests.log 2>&1
grep -v amdgpu.ids $O/tests.log | grep -E "passed|failed|rms|two-plane|two planes" | tail -12
cp pointnerf_amd/libpnerf_hip.so /tmp/shipped.so
for V in shipped "$@" shipped; do
  [ $V != shipped ] && cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  [ $V == shipped ] && cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
  timeout 300 python bench.py --cpu-rays 0 --steps 10 --wgrad-planes 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$V', 'planes 2:', round(d['value']), 'rays/s', round(d['ms_per_step'],2), 'ms  fwd', round(k['agg_forward']['ms_per_step'],2), 'bwd', round(k['agg_backward']['ms_per_step'],2), 'wgrad', round(k['wgrad']['ms_per_step'],2), 'reduce', round(k['wgrad_reduce']['ms_per_step'],2), 'loss', d['config']['final_loss'])"
done
cp /tmp/shipped.so pointnerf_amd/libpnerf_hip.so
