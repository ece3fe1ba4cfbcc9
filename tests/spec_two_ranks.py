"""Run by tests/test_gpu_train_steps.py under torch.distributed.run with 2 ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device).

The step enqueued before its host read (NeuralPointsRayMarching.render_dense) under data parallelism: the ranks see DIFFERENT valid-sample counts, so in the
second step rank 0 (whose arena was sized by a small first batch) drops its speculative result and runs the step again while rank 1 (whose first batch was
already full size) does not.  The redo issues no collective, so the ranks' collectives (loss denominators, gradient all-reduce) still pair up: the run must
finish (no hang), every rank must end with the same summed gradients, and those must equal the run with PNERF_SPECULATE=0.  One result per forward call
(the reference: models/neural_points_volumetric_model.py:252-329).  Prints one JSON line on rank 0."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(1, HERE)

from cases import build_case                                              # noqa: E402
from pointnerf_amd import dist as pdist, ops                              # noqa: E402
from pointnerf_amd import neural_points_volumetric_model as NM            # noqa: E402
from pointnerf_amd.neural_points import NeuralPoints                      # noqa: E402
from pointnerf_amd.point_aggregators import PointAggregator               # noqa: E402


def main():
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank = dist.get_rank()
    dev = torch.device("cuda:0")
    opt, xyz, attrs, inp, mlp = build_case("small_k8")
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    R = d["raydir"].shape[1]
    cut = lambda sl: dict(d, raydir=d["raydir"][:, sl].contiguous(), gt_image=d["gt_image"][:, sl].contiguous(), pixel_idx=d["pixel_idx"][:, sl].contiguous())
    mine = pdist.shard_slice(R)
    full = cut(mine)
    small = cut(slice(mine.start, mine.start + max((mine.stop - mine.start) // 4, 1)))
    batches = (small, full, full) if rank == 0 else (full, full, full)

    def run(speculate):
        NM.SPECULATE = speculate
        ops.ARENA.free = []
        agg = PointAggregator(opt).to(dev)
        agg.load_state_dict(mlp)
        agg.flatten_()
        npnt = NeuralPoints(32, xyz.shape[0], opt, dev)
        a = {k: v.to(dev) for k, v in attrs.items()}
        npnt.set_points(xyz.to(dev), a["points_embeding"], points_color=a["points_color"], points_dir=a["points_dir"], points_conf=a["points_conf"], parameter=True)
        model = NM.NeuralPointsRayMarching(aggregator=agg, neural_points=npnt, opt=opt)
        model.fused_zero_one = model.fused_color_loss = True
        mlp_params = list(agg.parameters())
        pt_params = [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]
        res, ahead, dropped = [], [], []
        for batch in batches:
            for p in mlp_params + pt_params:
                p.grad = None
            out = model(**batch)
            ahead.append(bool(model.last_stats["enqueued_before_host_read"]))
            dropped.append(bool(model.last_stats["speculative_result_dropped"]))
            loss = pdist.hot_path_loss(opt, out, batch["gt_image"])
            loss.backward()
            pdist.allreduce_grads(mlp_params, pt_params)
            tot = loss.detach().clone()
            dist.all_reduce(tot)
            res.append((float(tot), [p.grad.detach().cpu().clone() for p in mlp_params + pt_params]))
        return res, ahead, dropped

    ref, a0, d0 = run(False)
    got, a1, d1 = run(True)
    worst_l, worst_g = 0.0, 0.0
    for (l0, g0), (l1, g1) in zip(ref, got):
        worst_l = max(worst_l, abs(l0 - l1) / abs(l0))
        for x, y in zip(g0, g1):
            worst_g = max(worst_g, float((x - y).abs().max()) / max(float(x.abs().max()), 1e-12))
    # replicas: the summed gradients of the last step are the same tensors on both ranks
    sums = torch.tensor([float(g.double().sum()) for g in got[-1][1]], dtype=torch.float64)
    lo, hi = sums.clone(), sums.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    rec = dict(rank=rank, ahead_without=a0, ahead=a1, dropped=d1, worst_loss_rel=worst_l, worst_grad_rel=worst_g, replica_spread=float((hi - lo).abs().max()))
    recs = [None, None]
    dist.all_gather_object(recs, rec)
    if rank == 0:
        print(json.dumps(recs))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
