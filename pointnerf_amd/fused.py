"""The fused render step: query -> aggregator MLP -> ray-march, forward and backward, as ONE
torch.autograd.Function over the C-ABI library (no per-stage tensors cross the Python boundary).

This is what ``NeuralPointsRayMarching.forward`` (pointnerf_amd/neural_points_volumetric_model.py) runs; it
replaces the reference's ``neural_points(...) -> aggregator(...) -> ray_dist -> ray_march`` chain
(models/neural_points_volumetric_model.py:268-306).  Autograd sees four leaves: the flat MLP parameter vector and
the per-point tensors (embedding, conf, dir, colour); xyz gets no gradient (``xyz_grad=0`` in every reference
script, lego_cuda.sh: default of --xyz_grad in neural_points.py:132).
"""
import torch

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

    @staticmethod
    def forward(ctx, env, emb, conf, pdir, color, *mlp_params):
        # env: dict(cam, xyz, raydir, dense, R, SR, K, n_valid, flat, packed, train, layout)
        pts = ops.make_points(env["xyz"], emb.detach().reshape(-1, emb.shape[-1]), conf.detach().reshape(-1, 1),
                              pdir.detach().reshape(-1, 3), color.detach().reshape(-1, 3))
        fwd = ops.render_forward(env["cam"], pts, env["packed"], env["flat"], env["raydir"], env["dense"],
                                 env["R"], env["SR"], env["K"], env["n_valid"], env["train"])
        ctx.env, ctx.pts, ctx.fwd = env, pts, fwd
        ctx.shapes = (tuple(emb.shape), tuple(conf.shape), tuple(pdir.shape), tuple(color.shape))
        ctx.n_mlp = len(mlp_params)
        ctx.mark_non_differentiable(fwd["opacity"], fwd["bg_trans"], fwd["blend_w"], fwd["decoded"], fwd["weight"])
        return fwd["ray_color"], fwd["opacity"], fwd["bg_trans"], fwd["blend_w"], fwd["decoded"], fwd["weight"]

    @staticmethod
    def backward(ctx, g_color, *unused):
        env, fwd = ctx.env, ctx.fwd
        if not env["train"] or fwd["saved"] is None:
            raise RuntimeError("pointnerf_amd: backward through a render that was run with train=False")
        dev = g_color.device
        gflat = torch.zeros_like(env["flat"])
        names = ("points_embeding", "points_conf", "points_dir", "points_color")
        grads = {n: torch.zeros(shp, dtype=torch.float32, device=dev) for n, shp in zip(names, ctx.shapes)}
        if env["n_valid"] > 0:
            ops.render_backward(env["cam"], ctx.pts, env["packed"], env["flat"], env["raydir"], env["dense"], env["R"],
                                env["SR"], env["K"], env["n_valid"], fwd, g_color, gflat, grads)
        ops.ARENA.give(fwd["saved"])  # hand the activation arena back for the next step
        fwd["saved"] = None
        gm = tuple(gflat[o:o + n].view(shp) for (o, n, shp) in env["layout"])
        assert len(gm) == ctx.n_mlp
        return (None, grads["points_embeding"], grads["points_conf"], grads["points_dir"], grads["points_color"]) + gm
