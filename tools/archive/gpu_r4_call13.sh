#!/bin/bash
# round 4, call 13: weight fragments / bias requested in front of the barriers (shipped) against the build before (r3base): parity, then A/B
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r4c13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_backward.py tests/test_gpu_bench_config.py tests/test_gpu_reproducible.py tests/test_gpu_point_init.py -x -q > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -6
bash tools/gpu_ab.sh r3base "$@" 2>&1 | tee $O/ab.txt
cp gpurun_out/ab/*.json $O/ 2>/dev/null
for V in shipped r3base; do
  [ $V != shipped ] && cp pointnerf_amd/libpnerf_hip.so /tmp/s.so && cp tools/_build/$V.so pointnerf_amd/libpnerf_hip.so
  timeout 200 python bench.py --render-only --cpu-rays 0 --steps 10 > $O/render_$V.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/render_$V.json'));print('render-only $V', d['value'], d['ms_per_step'])"
  [ $V != shipped ] && cp /tmp/s.so pointnerf_amd/libpnerf_hip.so
done
