"""Ray-shard data parallelism for the hot path: one process per GPU, neural point cloud + MLP
replicated, the rays of a batch split contiguously by rank, gradients summed with one RCCL all-reduce per tensor
group (``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).

The reference has no multi-process path at all (its only multi-GPU code is ``torch.nn.DataParallel`` over a batch
dimension of size 1, models/neural_points_volumetric_model.py:165-168, which cannot split rays).

Collectives per step:
  * 2 floats  -- global element counts that normalise the two loss means (so that the SUM of per-rank gradients is
                 exactly the gradient of the single-process loss over the whole batch);
  * 1.37 MB   -- MLP gradients, flattened into one buffer (latency-bound, one all-reduce);
  * N x 39 f32 -- dense per-point gradients (312 MB at N = 2M), one all-reduce per tensor, in place.
xGMI is point-to-point (7 links per GPU): a ring all-reduce of S bytes moves 2(n-1)/n S over one link per GPU, so the
point-gradient payload is the scaling limiter; a touched-rows (sparse) exchange is the "next" step.
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_slice(n_rays_global, r=None, w=None):
    """Contiguous [begin, end) of rank r's rays; the remainder goes to the first ranks."""
    r = rank() if r is None else r
    w = world() if w is None else w
    base, rem = divmod(n_rays_global, w)
    b = r * base + min(r, rem)
    return slice(b, b + base + (1 if r < rem else 0))


def global_counts(*local_counts, device=None):
    """All-reduced element counts (float tensor) used as the denominators of the loss means."""
    t = torch.tensor([float(c) for c in local_counts], device=device)
    if world() > 1:
        dist.all_reduce(t)
    return t


def hot_path_loss(opt, out, gt_image, zero_epsilon=1e-3):
    """The lego script's training loss (models/base_rendering_model.py:543-551 ``ray_masked_coarse_raycolor`` x 1.0
    + 1e-6, and :630-641 ``zero_one`` on ``conf_coefficient`` x opt.zero_one_loss_weights[0]) with both means taken
    over the GLOBAL batch."""
    pred = out["coarse_raycolor"][0]
    gt = gt_image[0].index_select(0, out["_hit_index"]) if "_hit_index" in out else gt_image[0][out["ray_mask"][0] > 0]
    cc = out.get("conf_coefficient")
    n = global_counts(pred.numel(), cc.numel() if cc is not None else 0, device=pred.device)
    loss = ((pred - gt) ** 2).sum() / n[0].clamp(min=1.0) + 1e-6 / world()
    if cc is not None and "conf_coefficient" in opt.zero_one_loss_items:
        v = cc.clamp(zero_epsilon, 1 - zero_epsilon)
        loss = loss + (torch.log(v) + torch.log(1 - v)).sum() / n[1].clamp(min=1.0) * opt.zero_one_loss_weights[0]
    return loss


_COMM_STREAMS = {}


def allreduce_grads(mlp_params, point_params, ready_event=None, early_params=()):
    """Sum gradients over ranks, in place.  MLP: one flat bucket; points: one collective per tensor.

    ``ready_event`` / ``early_params`` (device tensors only): the gradients of ``early_params`` are complete once
    ``ready_event`` has fired (``FusedRender.point_grads_ready``: recorded by the library between the input-gradient kernels
    and the weight-gradient GEMMs), so their all-reduce is issued on a side stream behind that event and overlaps the ~20 ms
    of weight-gradient GEMMs instead of following them; the calling stream waits for the side stream before it returns.
    Only valid for parameters whose ``.grad`` IS the tensor the renderer's backward wrote (``zero_grad(set_to_none=True)`` and no
    other contribution in the graph: embedding, dir and colour -- not the confidences, which also receive the zero-one loss)."""
    if world() == 1:
        return
    # the side-stream all-reduce is only sound on the very tensor the renderer's backward wrote: if autograd cloned it (a non-stealable
    # layout, a pre-existing .grad, a hook) or added another contribution, p.grad is a different tensor that is completed on the main
    # stream AFTER the event, and the early reduction would race it -- such parameters take the ordinary path below
    from .fused import FusedRender
    early = [p for p in early_params if p.grad is not None and p.grad.is_cuda and p.grad.data_ptr() in FusedRender.point_grad_ptrs] \
        if ready_event is not None else []
    comm = None
    if early:
        dev = early[0].grad.device
        comm = _COMM_STREAMS.setdefault(dev, torch.cuda.Stream(device=dev))
        with torch.cuda.stream(comm):
            comm.wait_event(ready_event)
            for p in early:
                p.grad.record_stream(comm)
                dist.all_reduce(p.grad)
        point_params = [p for p in point_params if all(p is not q for q in early)]
    gs = [p.grad for p in mlp_params if p.grad is not None]
    if gs:
        flat = torch.cat([g.reshape(-1) for g in gs])
        dist.all_reduce(flat)
        o = 0
        for g in gs:
            g.copy_(flat[o:o + g.numel()].view_as(g)); o += g.numel()
    for p in point_params:
        if p.grad is not None:
            dist.all_reduce(p.grad)
    if comm is not None:
        torch.cuda.current_stream(comm.device).wait_stream(comm)
