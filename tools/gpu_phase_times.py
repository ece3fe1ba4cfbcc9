"""dev: wall time of each phase of a bench step (sync between phases)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointnerf_amd import config, ops
dev = torch.device('cuda:0')
opt = config.bench_lego_opt(is_train=0)
model = bench.build_model(opt, 2_000_000, dev)
agg, npnt = model.aggregator, model.neural_points
mlp_params = [p for p in agg.parameters()]
pt_params = [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]
o1 = torch.optim.Adam(mlp_params, lr=opt.lr); o2 = torch.optim.Adam(pt_params, lr=opt.plr)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for i in range(4):
    inp = bench.step_inputs(i, 0, 1, R, dev)
    t0 = T(); o1.zero_grad(set_to_none=True); o2.zero_grad(set_to_none=True)
    out = model(**inp); t1 = T()
    loss = bench.loss_fn(opt, out, inp, 1); t2 = T()
    loss.backward(); t3 = T()
    o1.step(); t4 = T(); o2.step(); t5 = T()
    print('step %d fwd %.1f loss %.1f bwd %.1f adam_mlp %.1f adam_pts %.1f ms | mem %.1f GB peak %.1f GB' % (
        i, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3, torch.cuda.memory_allocated()/2**30, torch.cuda.max_memory_allocated()/2**30))
    print('   emb grad absmax %.3e nonzero rows %d' % (npnt.points_embeding.grad.abs().max().item(), int((npnt.points_embeding.grad.abs().sum(-1) > 0).sum())))
