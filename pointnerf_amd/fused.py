"""The fused render step: query -> aggregator MLP -> ray-march, forward and backward, as ONE
torch.autograd.Function over the C-ABI library (no per-stage tensors cross the Python boundary).

This is what ``NeuralPointsRayMarching.forward`` (pointnerf_amd/neural_points_volumetric_model.py) runs; it
replaces the reference's ``neural_points(...) -> aggregator(...) -> ray_dist -> ray_march`` chain
(models/neural_points_volumetric_model.py:268-306).  Autograd sees four leaves: the flat MLP parameter vector and
the per-point tensors (embedding, conf, dir, colour); xyz gets no gradient (``xyz_grad=0`` in every reference
script, lego_cuda.sh: default of --xyz_grad in neural_points.py:132).
"""
import torch

from . import _lib as L
from . import ops


class MLPState:
    """Flat fp32 parameter vector + its MFMA-fragment image, re-packed lazily when the weights change."""

    def __init__(self, flat):
        self.flat = flat
        self.packed = None
        self._version = None

    def packed_image(self):
        # Always re-pack: optimizers update the parameters through their own views, which does not bump the
        # flat vector's version counter, and the pack kernel moves 2.7 MB (a few microseconds).
        self.packed = ops.pack_mlp(self.flat.detach(), self.packed)
        return self.packed


class FusedRender(torch.autograd.Function):
    """apply(env, emb, conf, dir, color, *mlp_params) -> (ray_color [R,3], opacity [R,SR], bg_trans [R],
    blend_w [R,SR], decoded [R,SR,4], weight [R,SR,K]); only ray_color carries gradient.

    ``mlp_params`` are the aggregator's nn.Parameters in ``pnerf_mlp_layout`` order; they are views of the flat
    vector ``env['flat']`` the kernels read, and are passed only so that autograd (and DDP-style hooks) see them:
    backward returns one gradient per parameter, each a view into a single flat gradient buffer."""

    # Class-level state describing the LATEST backward only: ONE render per optimisation step is assumed (every reference script; with
    # two renders in one graph the early all-reduce of dist.allreduce_grads falls back to the ordinary path for the earlier one, which
    # is safe -- its gradient tensors are not in point_grad_ptrs -- but not overlapped).
    point_grads_ready = None          # torch.cuda.Event of the latest backward (only when env["want_grad_event"])
    point_grad_ptrs = frozenset()     # data pointers of the point-gradient tensors the latest backward wrote
    point_grad_bucket = None          # (flat bucket, floats of its head [embedding | dir | colour], their data pointers)
    last_chunks = None                # (rays per chunk, rays) when the latest backward recomputed by ray chunks, else None

    @staticmethod
    def forward(ctx, env, emb, conf, pdir, color, *mlp_params):
        # env: dict(cam, xyz, raydir, dense, R, SR, K, n_valid, flat, packed, train, layout)
        # (the C structure holds raw pointers: keep the arrays it points to alive until the backward has read them)
        ctx.point_arrays = (emb.detach().reshape(-1, emb.shape[-1]), conf.detach().reshape(-1, 1), pdir.detach().reshape(-1, 3),
                            color.detach().reshape(-1, 3))
        pts = ops.make_points(env["xyz"], *ctx.point_arrays)
        # a step whose saved activations would exceed the arena budget runs its forward without saving anything; the backward then
        # re-runs the forward chunk of rays by chunk of rays (ops.arena_budget_bytes)
        from . import _lib as L
        ctx.recompute = bool(env["train"]) and L.lib().pnerf_agg_saved_bytes(env["n_valid"], env["K"]) > ops.arena_budget_bytes()
        fwd = ops.render_forward(env["cam"], pts, env["packed"], env["flat"], env["raydir"], env["dense"],
                                 env["R"], env["SR"], env["K"], env["n_valid"], env["train"] and not ctx.recompute)
        # only what the backward reads: ray_color must NOT be kept -- the returned tensor's grad_fn is this node, and node -> fwd -> ray_color ->
        # node is a reference cycle that only Python's cycle collector breaks: a training-mode forward that is never back-propagated (an evaluation
        # under enabled gradients) then holds its activation arena (tens of GB) until some later collection
        ctx.env, ctx.pts, ctx.fwd = env, pts, {k: fwd[k] for k in ("saved", "decoded", "weight", "opacity")}
        env["_saved"] = fwd["saved"]              # (a caller that drops this result without a backward hands the arena block back itself)
        ctx.shapes = (tuple(emb.shape), tuple(conf.shape), tuple(pdir.shape), tuple(color.shape))
        ctx.n_mlp = len(mlp_params)
        # env["zero_one_eps"] (training): the numerator of the zero-one regulariser on the hit rays' conf_coefficient is a SEVENTH, differentiable
        # output of this node (one pass over the dense neighbor table, pnerf_zero_one_forward_rays), so that its conf gradient can ride on
        # the conf atomics of this node's backward instead of repeating ~7 M atomics on the same addresses in a pass of its own
        ctx.zo = None
        zo_sum = None
        if env["train"] and env.get("zero_one_eps") is not None:
            lib = L.lib()
            dense, cflat = env["dense"], ctx.point_arrays[1].reshape(-1)
            R_, slots = env["R"], env["SR"] * env["K"]
            part = torch.empty(lib.pnerf_zero_one_blocks(R_ * 256), dtype=torch.float32, device=cflat.device)
            L.check(lib.pnerf_zero_one_forward_rays(ops._ptr(cflat), cflat.numel(), ops._ptr(dense["sample_pidx"]), ops._ptr(dense["ray_hit"]), R_, slots,
                                                    float(env["zero_one_eps"]), ops._ptr(part), ops._stream()), "pnerf_zero_one_forward_rays")
            zo_sum = part.sum()
            ctx.zo = float(env["zero_one_eps"])
        # ONE call (a second one replaces the set): every output the backward does not differentiate.  decoded / weight / opacity are kept in
        # ctx.fwd -- were they differentiable outputs, node -> ctx.fwd -> tensor -> grad_fn = node would be a reference cycle that holds the
        # activation arena of every training forward that is never back-propagated
        nondiff = [fwd["opacity"], fwd["bg_trans"], fwd["blend_w"], fwd["decoded"], fwd["weight"]]
        if zo_sum is None:
            zo_sum = torch.zeros((), dtype=torch.float32, device=fwd["ray_color"].device)
            nondiff.append(zo_sum)
        ctx.mark_non_differentiable(*nondiff)
        return fwd["ray_color"], fwd["opacity"], fwd["bg_trans"], fwd["blend_w"], fwd["decoded"], fwd["weight"], zo_sum

    @staticmethod
    def backward(ctx, g_color, *unused):
        env, fwd = ctx.env, ctx.fwd
        g_zo = unused[-1] if (ctx.zo is not None and unused) else None         # gradient of the seventh output (the zero-one numerator)
        if g_color is None:
            g_color = torch.zeros(env["R"], 3, dtype=torch.float32, device=env["raydir"].device)
        if not env["train"] or (fwd["saved"] is None and not ctx.recompute):
            raise RuntimeError("pointnerf_amd: backward through a render that was run with train=False")
        dev = g_color.device
        FusedRender.last_chunks = None
        gflat = torch.zeros_like(env["flat"])
        names = ("points_embeding", "points_conf", "points_dir", "points_color")
        # ONE zero-filled bucket for the four point-gradient tensors, [embedding | dir | colour | conf]: the three tensors that are final
        # at the library's ready event are contiguous at its head, so a data-parallel caller reduces them with one collective
        # (dist.allreduce_grads); the four gradients autograd receives are views of it
        order = (0, 2, 3, 1)
        sizes = [int(torch.Size(shp).numel()) for shp in ctx.shapes]
        offs, o = {}, 0
        for i in order:
            offs[i] = o
            o += (sizes[i] + 3) // 4 * 4                # every view starts on a 16-byte boundary
        bucket = torch.zeros(o, dtype=torch.float32, device=dev)
        grads = {n: bucket[offs[i]:offs[i] + sizes[i]].view(ctx.shapes[i]) for i, n in enumerate(names)}
        ev = None
        if env.get("want_grad_event"):            # data-parallel training: see pnerf_point_grads.ready_event
            ev = torch.cuda.Event()
            ev.record()                           # creates the hipEvent_t; re-recorded by the library between dgrad and wgrad
        zo = None if g_zo is None else (g_zo.detach().reshape(1).to(torch.float32).contiguous(), ctx.zo)
        if env["n_valid"] > 0 and ctx.recompute:
            FusedRender._backward_in_chunks(env, ctx.pts, g_color.contiguous().float(), gflat, grads, ev)
        elif env["n_valid"] > 0:
            ops.render_backward(env["cam"], ctx.pts, env["packed"], env["flat"], env["raydir"], env["dense"], env["R"],
                                env["SR"], env["K"], env["n_valid"], fwd, g_color, gflat, grads, ready_event=ev, zero_one=zo)
            zo = None
        if zo is not None:
            # (no valid sample at all, or the chunk-by-chunk recompute whose chunks carry their own counters: the regulariser's own pass)
            dense, cflat = env["dense"], ctx.point_arrays[1].reshape(-1)
            L.check(L.lib().pnerf_zero_one_backward_rays(ops._ptr(cflat), cflat.numel(), ops._ptr(dense["sample_pidx"]), ops._ptr(dense["ray_hit"]), env["R"],
                                                         env["SR"] * env["K"], zo[1], ops._ptr(zo[0]), ops._ptr(grads["points_conf"].reshape(-1)), ops._stream()),
                    "pnerf_zero_one_backward_rays")
        FusedRender.point_grads_ready = ev
        # what the early all-reduce may touch: exactly the tensors this backward wrote (dist.allreduce_grads checks p.grad against them)
        FusedRender.point_grad_ptrs = {grads[n].data_ptr() for n in names}
        FusedRender.point_grad_bucket = (bucket, offs[1], tuple(grads[n].data_ptr() for n in ("points_embeding", "points_dir", "points_color")))
        if fwd["saved"] is not None:
            ops.ARENA.give(fwd["saved"])  # hand the activation arena back for the next step
        fwd["saved"] = None
        env.pop("_saved", None)
        gm = tuple(gflat[o:o + n].view(shp) for (o, n, shp) in env["layout"])
        assert len(gm) == ctx.n_mlp
        # the graph node outlives this call for as long as the caller keeps the loss: release the step's big tensors (query
        # outputs, dense weights, packed points) now, so that the next step's allocations find them in the allocator's cache
        ctx.env = ctx.fwd = ctx.pts = ctx.point_arrays = None
        return (None, grads["points_embeding"], grads["points_conf"], grads["points_dir"], grads["points_color"]) + gm


def _backward_in_chunks(env, pts, g_color, gflat, grads, ev):
    """Backward of a render step whose saved activations do not fit the arena budget: for consecutive runs of rays, re-run the
    training forward (its saved activations within the budget) and the backward; gradients accumulate in the same buffers.
    One host synchronisation per chunk (its number of valid samples sizes its arena)."""
    from . import _lib as L
    lib = L.lib()
    dense, R, SR, K = env["dense"], env["R"], env["SR"], env["K"]
    budget = ops.arena_budget_bytes()
    per_ray = max(lib.pnerf_agg_saved_bytes(env["n_valid"], K) / max(R, 1), 1.0)
    step = max(int(0.8 * budget / per_ray), 1)
    todo = [(r0, min(r0 + step, R)) for r0 in range(0, R, step)]
    last, handed = None, False
    while todo:
        r0, r1 = todo.pop(0)
        nn = dense["sample_nn"][r0:r1]
        vlist, counters = ops.compact_valid(nn)
        n_c = int(counters[0].item())
        if n_c == 0:
            continue
        if lib.pnerf_agg_saved_bytes(n_c, K) > budget and r1 - r0 > 1:      # denser than the average: halve the run
            mid = (r0 + r1) // 2
            todo[:0] = [(r0, mid), (mid, r1)]
            continue
        sub = dict(sample_loc=dense["sample_loc"][r0:r1], sample_pidx=dense["sample_pidx"][r0:r1], sample_nn=nn, valid_list=vlist, counters=counters)
        rd = env["raydir"][r0:r1]
        f = ops.render_forward(env["cam"], pts, env["packed"], env["flat"], rd, sub, r1 - r0, SR, K, n_c, True)
        last = (r0, r1)
        # the library re-records the event between this chunk's input-gradient kernels and its weight-gradient GEMMs: only sound for
        # the chunk that is processed LAST (every earlier chunk's atomics must be behind it)
        handed = ev is not None and not todo
        ops.render_backward(env["cam"], pts, env["packed"], env["flat"], rd, sub, r1 - r0, SR, K, n_c, f, g_color[r0:r1], gflat, grads,
                            ready_event=ev if handed else None)
        ops.ARENA.give(f["saved"])
    if ev is not None and not handed:
        # the trailing run(s) of rays had no valid sample (or nothing was processed at all): the event still carries its record from
        # BEFORE the first chunk -- re-record it behind everything that was enqueued, or a data-parallel caller would all-reduce the
        # bucket while the chunks' atomics are still landing in it
        ev.record()
    FusedRender.last_chunks = None if last is None else (step, R)


FusedRender._backward_in_chunks = staticmethod(_backward_in_chunks)


class Aggregate(torch.autograd.Function):
    """The stand-alone aggregator (PointAggregator.forward, point_aggregators.py:727-814): gathered per-neighbor
    tensors in, (decoded [R,SR,4], weight [R,SR,K]) out.  The gathered arrays play the role of a point cloud of
    R*SR*K "points" indexed by slot, so the same kernels run; gradients come back per slot."""

    @staticmethod
    def forward(ctx, env, emb, conf, pdir, color, *mlp_params):
        # env: dict(cam, xyz_slots [N',3], xyz_pers [N',3], loc_w, loc_pers, raydir [R,3], pidx, nn, R, SR, K, flat, packed, train, layout)
        import ctypes
        from . import _lib as L
        lib = L.lib()
        dev = emb.device
        R, SR, K = env["R"], env["SR"], env["K"]
        nn = env["nn"]
        vlist, counters = ops.compact_valid(nn)
        n_valid = int(counters[0].item())
        # (the C structure holds raw pointers: the per-slot arrays must stay alive until the backward has read them)
        slot_arrays = (emb.detach().reshape(-1, emb.shape[-1]).contiguous(), conf.detach().reshape(-1, 1).contiguous(),
                       pdir.detach().reshape(-1, 3).contiguous(), color.detach().reshape(-1, 3).contiguous())
        pts = ops.make_points(env["xyz_slots"], *slot_arrays)
        f32 = dict(dtype=torch.float32, device=dev)
        decoded, weight = torch.empty(R, SR, 4, **f32), torch.empty(R, SR, K, **f32)
        saved = ws = None
        nw = 0
        if env["train"]:
            saved = ops.ARENA.take(lib.pnerf_agg_saved_bytes(n_valid, K), dev)
        else:
            nw = lib.pnerf_agg_workspace_bytes(n_valid, K)
            ws = torch.empty(nw, dtype=torch.uint8, device=dev)
        L.check(lib.pnerf_agg_forward(ctypes.byref(env["cam"]), ctypes.byref(pts), ops._ptr(env["packed"]), ops._ptr(env["flat"]),
                                      ops._ptr(env["raydir"]), ops._ptr(env["loc_w"]), ops._ptr(env["xyz_pers"]), ops._ptr(env["loc_pers"]),
                                      ops._ptr(env["pidx"]), ops._ptr(vlist), ops._ptr(counters), R, SR, K, ops._ptr(decoded), ops._ptr(weight),
                                      ops._ptr(saved), n_valid, ops._ptr(ws), nw, ops._stream()), "pnerf_agg_forward")
        ctx.env, ctx.pts, ctx.keep = env, pts, (vlist, counters, saved, decoded, weight, n_valid)
        ctx.slot_arrays = slot_arrays
        ctx.shapes = (tuple(emb.shape), tuple(conf.shape), tuple(pdir.shape), tuple(color.shape))
        ctx.mark_non_differentiable(weight)
        return decoded, weight

    @staticmethod
    def backward(ctx, g_decoded, *unused):
        import ctypes
        from . import _lib as L
        lib = L.lib()
        env = ctx.env
        vlist, counters, saved, decoded, weight, n_valid = ctx.keep
        if saved is None:
            raise RuntimeError("pointnerf_amd: backward through an aggregator call that was run with train=False")
        dev = g_decoded.device
        R, SR, K = env["R"], env["SR"], env["K"]
        gflat = torch.zeros_like(env["flat"])
        names = ("embedding", "conf", "dir", "color")
        grads = [torch.zeros(shp, dtype=torch.float32, device=dev) for shp in ctx.shapes]
        pg = L.PointGrads()
        pg.embedding, pg.conf, pg.dir, pg.color = [g.data_ptr() for g in grads]
        nws = lib.pnerf_render_backward_workspace_bytes(0, 1)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        gd = g_decoded.reshape(R, SR, 4).contiguous().float()
        if n_valid > 0:
            L.check(lib.pnerf_agg_backward(ctypes.byref(env["cam"]), ctypes.byref(ctx.pts), ops._ptr(env["packed"]), ops._ptr(env["flat"]),
                                           ops._ptr(env["raydir"]), ops._ptr(env["loc_w"]), ops._ptr(env["pidx"]), ops._ptr(vlist),
                                           ops._ptr(counters), R, SR, K, n_valid, ops._ptr(decoded), ops._ptr(weight), ops._ptr(gd),
                                           ops._ptr(saved), ops._ptr(gflat), ctypes.byref(pg), ops._ptr(ws), nws, ops._stream()),
                    "pnerf_agg_backward")
        ops.ARENA.give(saved)
        ctx.keep = ctx.slot_arrays = None
        gm = tuple(gflat[o:o + n].view(shp) for (o, n, shp) in env["layout"])
        return (None,) + tuple(grads) + gm
