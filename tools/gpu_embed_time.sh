#!/bin/bash
# round 5, call 21: f4 extract_2d / query_embedding on the device + timing at 2 M points
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
O=gpurun_out/r5c21; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_point_init.py -q -s > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -15
timeout 300 python - > $O/embed_time.json 2>$O/embed_time.err <<'PY'
import json, types, torch, sys
from shell_fakes import embed_inputs
from pointnerf_amd.mvs_points_model import MvsPointsModel
dev="cuda:0"
inp=embed_inputs(seed=9,n=2_000_000,HD=512,WD=640,focal=520.0)
res={}
# the candidates of the reference come from back-projected depth maps, i.e. in pixel-raster order of their view; the fixture's are in random order
xyz=inp["cam_xyz"]
u=(xyz[0,:,0]/xyz[0,:,2]*520.0+320.0).round().clamp(0,639).long(); v=(xyz[0,:,1]/xyz[0,:,2]*520.0+256.0).round().clamp(0,511).long()
raster=torch.argsort(v*640+u)
for order,pts in (("random_order",xyz),("raster_order",xyz[:,raster].contiguous())):
 for occ in (0,1):
    m=MvsPointsModel(types.SimpleNamespace(depth_occ=occ,ref_vid=0,shading_feature_mlp_layer0=0))
    feats=[f.to(dev) for f in inp["img_feats"]]
    a=(feats,[0,1,2],[0,1,2,3],inp["intrinsics"].to(dev),inp["c2ws"].to(dev),inp["w2cs"].to(dev),pts.to(dev),inp["HD"],inp["WD"])
    for _ in range(3): m.extract_2d(*a,cam_vid=0)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): m.extract_2d(*a,cam_vid=0)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    byt=2_000_000*(12+16*3+4*(168+9))
    res["%s_depth_occ_%d"%(order,occ)]={"ms":ms,"algorithmic_GB":byt/1e9,"GBps":byt/ms/1e6}
print(json.dumps({"what":"MvsPointsModel.extract_2d, 2 M points x 3 views x (image + 3 pyramid levels of 8/16/32 channels), 512x640 maps","result":res}))
PY
cat $O/embed_time.json; tail -3 $O/embed_time.err
