"""Ray-shard data parallelism for the hot path: one process per GPU, neural point cloud + MLP
replicated, the rays of a batch split contiguously by rank, gradients summed with one RCCL all-reduce per tensor
group (``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).

The reference has no multi-process path at all (its only multi-GPU code is ``torch.nn.DataParallel`` over a batch
dimension of size 1, models/neural_points_volumetric_model.py:165-168, which cannot split rays).

Collectives per step:
  * 2 floats  -- global element counts that normalise the two loss means (so that the SUM of per-rank gradients is
                 exactly the gradient of the single-process loss over the whole batch);
  * 1.37 MB   -- MLP gradients, flattened into one buffer (latency-bound, one all-reduce);
  * N x 39 f32 -- dense per-point gradients (312 MB at N = 2M): embedding + dir + colour (38 N) as ONE in-place all-reduce of
                 the renderer's gradient bucket, issued on a side stream behind the library's "point gradients final" event so
                 that it overlaps the weight-gradient GEMMs; the confidences (N, they also receive the zero-one loss) afterwards.
xGMI is point-to-point (7 links per GPU): a ring all-reduce of S bytes moves 2(n-1)/n S over one link per GPU, so the
point-gradient payload is the scaling limiter.  For large clouds (a rank's rays touch a few percent of 6-20 M points) the
dense exchange is replaced by ``sparse_allreduce_rows``: all-gather of (row id, row) of the touched rows only, added in rank
order on every rank (bitwise identical replicas, same sum as the dense all-reduce up to fp32 association).
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# Test / bring-up switch (bench.py --force-collectives): with an initialised process group of ONE rank, run every collective of the step anyway (each is
# then the identity).  RCCL refuses two ranks on one device, so on a 1-GPU box this is the only way the step's collective calls -- the side-stream
# all-reduce of the gradient bucket behind the library's event, the in-place reduce-scatter / all-gather of ZeRO-1, all_gather_into_tensor of the sparse
# exchange -- execute on the real backend before an 8-GPU node does.
FORCE_COLLECTIVES = False


def active():
    """True when the step's collectives have to run: more than one rank, or the bring-up switch with an initialised group"""
    return world() > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_slice(n_rays_global, r=None, w=None):
    """Contiguous [begin, end) of rank r's rays; the remainder goes to the first ranks."""
    r = rank() if r is None else r
    w = world() if w is None else w
    base, rem = divmod(n_rays_global, w)
    b = r * base + min(r, rem)
    return slice(b, b + base + (1 if r < rem else 0))


def at_least_one(n):
    """a loss denominator: ``max(n, 1)`` for the plain number a single process gets from ``global_counts`` and for the all-reduced tensor element"""
    return n.clamp(min=1.0) if isinstance(n, torch.Tensor) else max(float(n), 1.0)


def global_counts(*local_counts, device=None):
    """All-reduced element counts used as the denominators of the loss means (a float tensor under torch.distributed; plain numbers in a
    single process).  No ``torch.tensor(list, device=...)``: a host -> device copy from pageable memory is a BLOCKING call on the stream --
    round 4 found it in the kernel trace as the point where the host, until then a whole forward ahead of the device, waited for the device
    and then paced every launch of the loss and of the backward's head (0.25 .. 0.5 ms of idle device per step)."""
    if not active():
        return [float(c) for c in local_counts]
    t = torch.zeros(len(local_counts), dtype=torch.float32, device=device)
    for i, c in enumerate(local_counts):
        t[i].fill_(float(c))                 # (a fill kernel with the number as its argument: asynchronous)
    dist.all_reduce(t)
    return t


def hot_path_loss(opt, out, gt_image, zero_epsilon=1e-3):
    """The lego script's training loss (models/base_rendering_model.py:543-551 ``ray_masked_coarse_raycolor`` x 1.0
    + 1e-6, and :630-641 ``zero_one`` on ``conf_coefficient`` x opt.zero_one_loss_weights[0]) with both means taken
    over the GLOBAL batch."""
    cc, zo, dc, zs = out.get("conf_coefficient"), out.get("_zero_one"), out.get("_dense_color"), out.get("_zero_one_sum")
    n_cc = cc.numel() if cc is not None else (zo[3] if zo is not None else (zs[1] if zs is not None else 0))
    if dc is not None:                # the renderer handed out (dense ray colours, hit flags, number of hit rays): one fused pass, no compaction
        from . import ops
        n = global_counts(3 * dc[2], n_cc, device=dc[0].device)
        loss = ops.color_loss_sum_rays(dc[0], gt_image[0], dc[1]) / at_least_one(n[0]) + 1e-6 / world()
    else:
        pred = out["coarse_raycolor"][0]
        gt = gt_image[0].index_select(0, out["_hit_index"]) if "_hit_index" in out else gt_image[0][out["ray_mask"][0] > 0]
        n = global_counts(pred.numel(), n_cc, device=pred.device)
        loss = ((pred - gt) ** 2).sum() / at_least_one(n[0]) + 1e-6 / world()
    if "conf_coefficient" in opt.zero_one_loss_items:
        if zs is not None:            # the render node computed the numerator itself (its conf gradient rides on the node's backward)
            loss = loss + zs[0] / at_least_one(n[1]) * opt.zero_one_loss_weights[0]
        elif zo is not None:          # the renderer handed out (points_conf, neighbor table) instead of the tensor: one fused pass
            from . import ops
            loss = loss + ops.zero_one_conf_sum_rays(zo[0], zo[1], zo[2], zero_epsilon) / at_least_one(n[1]) * opt.zero_one_loss_weights[0]
        elif cc is not None:
            v = cc.clamp(zero_epsilon, 1 - zero_epsilon)
            loss = loss + (torch.log(v) + torch.log(1 - v)).sum() / at_least_one(n[1]) * opt.zero_one_loss_weights[0]
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
    if not active():
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
        # embedding, dir and colour gradients are the head of one bucket (fused.FusedRender.backward): ONE collective when all three
        # are reduced early, else one per tensor
        bk = FusedRender.point_grad_bucket
        whole = bk is not None and sorted(p.grad.data_ptr() for p in early) == sorted(bk[2])
        with torch.cuda.stream(comm):
            comm.wait_event(ready_event)
            if whole:
                bk[0].record_stream(comm)
                dist.all_reduce(bk[0][:bk[1]])
            else:
                for p in early:
                    p.grad.record_stream(comm)
                    dist.all_reduce(p.grad)
        point_params = [p for p in point_params if all(p is not q for q in early)]
    gs = [p.grad for p in mlp_params if p.grad is not None]
    if gs:
        # the renderer's backward returns the MLP gradients as views of ONE flat buffer in parameter order (fused.FusedRender.backward:
        # gflat): when autograd handed them through unchanged they are contiguous in memory and the all-reduce runs in place on that
        # buffer -- no torch.cat, no copy back
        esz = gs[0].element_size()
        contiguous = all(g.is_contiguous() and g.dtype == gs[0].dtype for g in gs) and \
            all(gs[i].data_ptr() + gs[i].numel() * esz == gs[i + 1].data_ptr() for i in range(len(gs) - 1))
        st0 = gs[0].untyped_storage().data_ptr()
        if contiguous and all(g.untyped_storage().data_ptr() == st0 for g in gs):
            dist.all_reduce(gs[0].as_strided((sum(g.numel() for g in gs),), (1,), gs[0].storage_offset()))
        else:
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


def touched_flags(pidx, n_points):
    """[n_points] 0/1 int32 flags of the points a rank's gradients can be non-zero for (``pidx`` int tensor, -1 = empty slot).
    Row 0 is touched whenever ANY slot is empty: empty slots read point 0 like the reference (neural_points.py:709 clamps the index),
    and the zero-one regulariser differentiates through that read into points_conf[0] (k_zero_one_backward / the unfused gather).
    Device tables: one library pass over the int32 table (ops.touched_flags); CPU tensors (gloo tests): torch."""
    if n_points <= 0:
        return torch.zeros(0, dtype=torch.int32, device=pidx.device)
    if pidx.is_cuda and pidx.dtype == torch.int32:
        from . import ops
        return ops.touched_flags(pidx.contiguous(), n_points)
    flag = torch.zeros(n_points + 1, dtype=torch.int32, device=pidx.device)
    flag[pidx.reshape(-1).long() + 1] = 1             # slot 0 collects the -1 entries
    flag[1] |= flag[0]
    return flag[1:]


def touched_rows(pidx, n_points):
    """Sorted ids of the touched points (see touched_flags).  One host synchronisation (nonzero): the step-time callers use
    ``plan_sparse_exchange`` instead, which folds the count into the step's one existing host read."""
    return torch.nonzero(touched_flags(pidx, n_points)).reshape(-1)


def plan_sparse_exchange(pidx, n_points, group=None):
    """Everything ``sparse_allreduce_rows`` needs, computed right after the query and BEFORE the step's one host read: returns
    (ids [n_points] int32 device tensor whose first ``count`` entries are the sorted touched rows, counts [2] int64 DEVICE tensor =
    (this rank's count, max count over the ranks)).  The caller reads ``counts`` together with the query counters (one synchronisation
    for the step) and passes (ids, count, cap) on.  The compaction runs in libpnerf_hip.so on device tensors; gloo / CPU tensors (tests)
    take torch.nonzero."""
    flags = touched_flags(pidx, n_points)
    if flags.is_cuda:
        from . import ops
        ids, counters = ops.compact_valid(flags)                  # ascending indices of the flags > 0, count in counters[0]
        cnt = counters[:1].to(torch.int64)
    else:
        nz = torch.nonzero(flags).reshape(-1).to(torch.int32)
        ids = torch.zeros(n_points, dtype=torch.int32)
        ids[:nz.numel()] = nz
        cnt = torch.tensor([nz.numel()], dtype=torch.int64)
    cap = cnt.clone()
    if active():
        dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=group)
    return ids, torch.cat([cnt, cap])


def sparse_allreduce_rows(grads, touched, group=None, cap=None):
    """Sum over ranks of dense per-point gradients ``grads`` (list of [N, c_i] tensors, each rank's contribution zero outside its own
    ``touched`` rows) by exchanging only touched rows: all-gather (ids padded to the largest count with distinct rows whose values are zero, rows [cap, sum c_i]),
    then every rank zeroes its touched rows and adds the blocks of ALL ranks in rank order -- the same additions in the same order
    everywhere, so the replicas stay bitwise identical (a dense ring all-reduce has that property by construction).
    Bytes received per rank: W * cap * (4 + 4 sum c_i), against 2 (W - 1) / W * 4 N sum c_i for the dense ring.
    ``cap`` = the largest touched count over the ranks when the caller already knows it (``plan_sparse_exchange``: no host
    synchronisation in here then)."""
    W = dist.get_world_size(group)
    if W == 1 and not FORCE_COLLECTIVES:
        return
    dev = grads[0].device
    assert all(g.is_contiguous() and g.dim() >= 2 for g in grads)
    flat = [g.view(-1, g.shape[-1]) for g in grads]         # views: the additions below land in the callers' tensors
    cols = [g.shape[1] for g in flat]
    if cap is None:                                        # (callers without a plan: one collective + one host read here)
        cap = torch.tensor([touched.numel()], dtype=torch.int64, device=dev)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=group)
        cap = int(cap.item())
    cap = max(int(cap), 1)
    # padding: all-zero rows -- adding 0.0 to a row changes nothing, so no rank has to mask (a boolean-mask index is a device -> host read
    # per rank block: W stalls inside the exchange) and every rank still performs the same additions in the same order.  The padded entries
    # name DISTINCT rows (entry j -> row j mod N): with one common row, a rank block with few touched rows issued (cap - count) x 39
    # same-address atomics of index_add_ onto row 0's cache lines
    ids = torch.arange(cap, dtype=torch.int64, device=dev).remainder_(max(int(flat[0].shape[0]), 1))
    ids[:touched.numel()] = touched
    rows = torch.zeros(cap, sum(cols), dtype=torch.float32, device=dev)
    if touched.numel():
        rows[:touched.numel()] = torch.cat([g.index_select(0, touched) for g in flat], dim=1)
    all_ids = torch.empty(W * cap, dtype=torch.int64, device=dev)
    all_rows = torch.empty(W * cap, sum(cols), dtype=torch.float32, device=dev)
    if dist.get_backend(group) == "gloo":                  # (CPU tests)
        dist.all_gather(list(all_ids.view(W, cap).unbind(0)), ids, group=group)
        dist.all_gather(list(all_rows.view(W, cap, -1).unbind(0)), rows, group=group)
    else:
        dist.all_gather_into_tensor(all_ids, ids, group=group)
        dist.all_gather_into_tensor(all_rows, rows, group=group)
    if touched.numel():
        for g in flat:
            g.index_fill_(0, touched, 0.0)
    for r in range(W):
        i = all_ids[r * cap:(r + 1) * cap]
        blk = all_rows[r * cap:(r + 1) * cap]
        o = 0
        for g, c in zip(flat, cols):
            g.index_add_(0, i, blk[:, o:o + c])
            o += c
