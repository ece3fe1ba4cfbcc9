"""The model shell the reference's run scripts drive (SURVEY.md 8b "model" row): ``create_model(opt)`` returns an object with
the methods ``run/train_ft.py`` and ``run/test_ft.py`` call around the hot path --

    setup / set_input / optimize_parameters / test / forward / eval / train / get_current_losses / get_current_visuals /
    save_networks / load_networks / update_learning_rate / setup_optimizer / init_scheduler / clean_optimizer /
    clean_scheduler / clean_optimizer_scheduler / reset_optimizer / reset_scheduler / prune_points / grow_points /
    set_points / reset_ray_miss_ranking / top_ray_miss_loss / top_ray_miss_ids / cleanup

(run/train_ft.py:55-76,151,217-218,302-303,478,634,756-765,784,836-840,872-873,937-943,964).  It restates, for the
point-nerf-only mode the per-scene scripts use (``--mode 2``), the class chain
``MvsPointsVolumetricModel`` (models/mvs_points_volumetric_model.py:14-344) ->
``NeuralPointsVolumetricModel`` (models/neural_points_volumetric_model.py:7-220) ->
``BaseRenderingModel`` (models/base_rendering_model.py:19-674) -> ``BaseModel`` (models/base_model.py:7-156).

What is different in HOW:
  * the network is ``pointnerf_amd.NeuralPointsRayMarching`` (libpnerf_hip.so) and is not wrapped in ``DataParallel`` (the
    reference's only multi-GPU device, over a batch of 1); under ``torch.distributed`` the rays are sharded by rank, the two
    loss means are normalised by GLOBAL counts and gradients are summed over ranks before the step (``dist.py``);
  * the two Adam instances are ``optim.FusedAdam`` (one HIP pass per tensor, torch's state layout so optimizer
    checkpoints interchange);
  * the masked colour loss is taken on the compact hit-ray tensor the renderer returns instead of ``masked_select``-ing it
    back out of the scattered [1,R,3] image (same elements in the same order).
The MVSNet branch (``mode != 2``: ``net_mvs``, ``gen_points``, ``query_embedding``, ``set_bg``) is out of scope
and raises.
"""
import os

import numpy as np
import torch
from torch import nn
from torch.optim import lr_scheduler

from . import dist as pdist
from .neural_points import NeuralPoints
from .neural_points_volumetric_model import NeuralPointsRayMarching, fill_invalid
from .optim import FusedAdam
from .point_aggregators import PointAggregator


def get_scheduler(optimizer, opt):
    """models/helpers/networks.py:41-68 (the policies that function can actually construct)."""
    if opt.lr_policy == "lambda":
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda it: 1.0 - max(0, it - opt.niter) / float(opt.niter_decay + 1))
    if opt.lr_policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == "plateau":
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.2, threshold=0.01, patience=5)
    if opt.lr_policy == "iter_exponential_decay":
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda it: pow(opt.lr_decay_exp, it / opt.lr_decay_iters))
    raise NotImplementedError("learning rate policy [%s] is not implemented" % opt.lr_policy)


def _as_list(v):
    if v is None:
        return []
    if isinstance(v, str):
        return v.split()
    return list(v)


def _broadcast_weights(weights, items, what):
    """base_rendering_model.py:226-266: one weight is broadcast over the items, otherwise the lengths must agree."""
    weights = [float(w) for w in np.asarray(weights, dtype=np.float64).reshape(-1)]
    if len(weights) == 1 and len(items) > 1:
        weights = weights * len(items)
    if len(items) and len(weights) != len(items):
        raise ValueError("%s_loss_weights does not match loss items" % what)
    return weights


class MvsPointsVolumetricModel:

    def name(self):
        return self.__class__.__name__

    # ------------------------------------------------------------------ construction (base_model.py:14-25, base_rendering_model.py:361-385)
    def initialize(self, opt):
        if getattr(opt, "mode", 2) != 2:
            raise NotImplementedError("only --mode 2 (point-nerf only) is built; the MVSNet initialiser is out of scope")
        self.opt = opt
        self.gpu_ids = _as_list(getattr(opt, "gpu_ids", [0]))
        self.is_train = opt.is_train
        self.device = torch.device("cuda:{}".format(self.gpu_ids[0])) if self.gpu_ids else torch.device("cpu")
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names, self.model_names, self.visual_names = [], [], []
        self.optimizers, self.schedulers = [], []
        self.optimizer = self.neural_point_optimizer = None
        self.output = self.input = self.gt_image = self.gt_depth = self.gt_mask = None
        self.top_ray_miss_ids = self.top_ray_miss_loss = None
        self.loss_total = None
        self._raw = None
        self.check_setup_loss(opt)
        if len(self.loss_names) == 1 and opt.is_train:
            raise NotImplementedError("Requiring losses to train")
        self.check_setup_visuals(opt)
        self.create_network_models(opt)
        if self.is_train:
            self.setup_optimizer(opt)

    def check_setup_loss(self, opt):
        """base_rendering_model.py:221-274; the colour default of neural_points_volumetric_model.py:73-77."""
        self.loss_names = ["total"]
        opt.color_loss_items = _as_list(getattr(opt, "color_loss_items", None))
        if not opt.color_loss_items:
            opt.color_loss_items = ["coarse_raycolor"]
        opt.color_loss_weights = _broadcast_weights(getattr(opt, "color_loss_weights", [1.0]), opt.color_loss_items, "color")
        self.loss_names += opt.color_loss_items
        for what in ("depth", "zero_one", "bg", "l2_size"):
            items = _as_list(getattr(opt, what + "_loss_items", None))
            setattr(opt, what + "_loss_items", items)
            setattr(opt, what + "_loss_weights", _broadcast_weights(getattr(opt, what + "_loss_weights", [1.0]), items, what))
        if opt.depth_loss_items or opt.bg_loss_items:
            raise NotImplementedError("depth / bg losses need compute_depth and gt masks, which no point-nerf script enables")
        self.loss_names += opt.zero_one_loss_items
        if opt.sparse_loss_weight > 0:
            self.loss_names += ["sparse"]
        self.l2loss = torch.nn.MSELoss()
        self.l1loss = torch.nn.L1Loss()

    def check_setup_visuals(self, opt):
        """base_rendering_model.py:276-286, defaults of neural_points_volumetric_model.py:79-82."""
        if getattr(opt, "visual_items", None) is None:
            opt.visual_items = ["gt_image", "coarse_raycolor", "queried_shading"]
            self.visual_names += opt.visual_items + _as_list(getattr(opt, "visual_items_additional", None))
        else:
            self.visual_names += _as_list(opt.visual_items)

    def create_network_models(self, opt):
        """neural_points_volumetric_model.py:133-168: aggregator, neural points (restored from
        ``<checkpoints_dir>/<name>/<resume_iter>_net_ray_marching.pth`` when that file exists), ray-marching network."""
        self.aggregator = PointAggregator(opt).to(self.device)
        ckpt = os.path.join(opt.checkpoints_dir, opt.name, "{}_net_ray_marching.pth".format(opt.resume_iter))
        ckpt = ckpt if os.path.isfile(ckpt) else None
        if opt.num_point > 0:
            self.neural_points = NeuralPoints(opt.point_features_dim, opt.num_point, opt, self.device, checkpoint=ckpt,
                                              feature_init_method=opt.feature_init_method, reg_weight=0., feedforward=opt.feedforward)
        else:
            self.neural_points = None
        self.net_ray_marching = NeuralPointsRayMarching(aggregator=self.aggregator, neural_points=self.neural_points, opt=opt,
                                                        num_pos_freqs=getattr(opt, "num_pos_freqs", 0),
                                                        num_viewdir_freqs=getattr(opt, "num_viewdir_freqs", 0)).to(self.device)
        if self.device.type == "cuda":
            self.aggregator.flatten_()
            self.net_ray_marching.fused_zero_one = True      # compute_losses takes the fused zero-one pass (ops.ZeroOneConf)
            # the colour loss over the renderer's DENSE ray colours (ops.ColorLossRays; no compaction of the hit rays, no scatter back): taken when
            # every colour-loss item that carries a gradient is a ray_masked one -- the lego script's items are (ray_masked 1.0, ray_miss 0.0,
            # full image 0.0); ray_miss predicts the constant background and has no gradient, a full-image item with a non-zero weight would
            # need the gradient of the filled image and keeps the compacted form
            # On that path the rendered outputs in self.output (coarse_raycolor, coarse_point_opacity, coarse_is_background) are DETACHED -- the
            # gradient flows through ops.color_loss_sum_rays only -- so it is not taken when anything else could want a gradient through a rendered
            # tensor: an l2-size item, the sparse loss (it reads the opacities), or opt.fused_color_loss = 0 (the opt-out for subclasses / hooks that
            # build their own loss from model.output; the reference's outputs are differentiable).
            items = list(zip(getattr(opt, "color_loss_items", []), getattr(opt, "color_loss_weights", [])))
            others = bool(getattr(opt, "l2_size_loss_items", None)) or float(getattr(opt, "sparse_loss_weight", 0) or 0) > 0
            self.net_ray_marching.fused_color_loss = bool(items) and not others and int(getattr(opt, "fused_color_loss", 1)) != 0 and all(
                n == "ray_masked_coarse_raycolor" or n.startswith("ray_miss") or float(w) == 0.0 for n, w in items)
        self.model_names = ["ray_marching"]

    def get_networks(self):
        return [getattr(self, "net_" + n) for n in self.model_names]

    def setup(self, opt, train_len=None):
        """base_model.py:34-44 + mvs_points_volumetric_model.py:158-165."""
        if self.is_train:
            self.schedulers = [get_scheduler(o, opt) for o in self.optimizers]
        if not self.is_train or opt.resume_dir:
            self.load_networks(opt.resume_iter)
        self.print_networks(getattr(opt, "verbose", 0))
        if opt.prob_freq > 0 and train_len is not None and opt.prob_num_step > 1:
            self.num_probe = train_len // opt.prob_num_step
            self.reset_ray_miss_ranking()
        elif opt.prob_freq > 0 and train_len is not None and opt.prob_num_step == 1:
            self.top_ray_miss_loss = torch.zeros([1], dtype=torch.float32, device=self.device)

    def print_networks(self, verbose):
        for name, net in zip(self.model_names, self.get_networks()):
            n = sum(p.numel() for p in net.parameters())
            if verbose:
                print(net)
            print("[Network {}] Total number of parameters: {:.3f}M".format(name, n / 1e6))

    def eval(self):
        for net in self.get_networks():
            net.eval()

    def train(self):
        for net in self.get_networks():
            net.train()

    # ------------------------------------------------------------------ optimizers / schedulers (mvs_points_volumetric_model.py:47-95,196-233)
    def setup_optimizer(self, opt):
        named = list(self.net_ray_marching.named_parameters())
        self.net_params = [p for n, p in named if not n.startswith("neural_points")]
        self.neural_params = [p for n, p in named if n.startswith("neural_points")]
        self.mvs_params = []
        self.optimizers = []
        self.optimizer = self.neural_point_optimizer = None
        if self.net_params:
            self.optimizer = FusedAdam(self.net_params, lr=opt.lr, betas=(0.9, 0.999))
            self.optimizers.append(self.optimizer)
        if self.neural_params:
            self.neural_point_optimizer = FusedAdam(self.neural_params, lr=opt.plr, betas=(0.9, 0.999))
            self.optimizers.append(self.neural_point_optimizer)

    def reset_optimizer(self, opt):
        self.clean_optimizer()
        self.setup_optimizer(opt)

    def clean_optimizer(self):
        self.optimizers = []
        self.net_params, self.neural_params, self.mvs_params = [], [], []
        self.optimizer = self.neural_point_optimizer = None

    def clean_scheduler(self):
        self.schedulers = []

    def clean_optimizer_scheduler(self):
        self.clean_optimizer()
        self.clean_scheduler()

    def init_scheduler(self, total_steps, opt):
        """Fresh schedulers fast-forwarded to ``total_steps`` (what the train loop does after prune / grow rebuilt the
        optimizers, run/train_ft.py:836-840)."""
        self.schedulers = [get_scheduler(o, opt) for o in self.optimizers]
        for s in self.schedulers:
            for _ in range(int(total_steps)):
                s.step()

    reset_scheduler = init_scheduler

    def update_learning_rate(self, **kwargs):
        """base_model.py:143-156."""
        for s in self.schedulers:
            s.step()
        for i, o in enumerate(self.optimizers):
            lr = o.param_groups[0]["lr"]
            opt = kwargs.get("opt")
            if opt is None or not opt.lr_policy.startswith("iter") or \
                    ("total_steps" in kwargs and kwargs["total_steps"] % opt.print_freq == 0):
                print("optimizer {}, learning rate = {:.7f}".format(i + 1, lr))

    # ------------------------------------------------------------------ the step
    def set_input(self, input):
        """base_rendering_model.py:387-405."""
        self.input = input
        for k, v in self.input.items():
            if isinstance(v, torch.Tensor):
                self.input[k] = v.to(self.device)
        self.gt_image = self.input["gt_image"] if "gt_image" in input else None
        self.gt_depth = self.input["gt_depth"] if "gt_depth" in input else None
        self.gt_mask = self.input["gt_mask"] if "gt_mask" in input else None

    def run_network_models(self):
        """neural_points_volumetric_model.py:84-85."""
        self._raw = self.net_ray_marching(**self.input)
        return fill_invalid(self._raw, self.input.get("bg_color"), bg_ray=self.input.get("bg_ray"), prob=getattr(self.opt, "prob", 0))

    def set_visuals(self):
        for k, v in self.output.items():
            if k in self.visual_names:
                setattr(self, k, v)
        if "coarse_raycolor" not in self.visual_names:
            self.coarse_raycolor = self.output["coarse_raycolor"]

    def forward(self):
        self.output = self.run_network_models()
        self.set_visuals()
        if not self.opt.no_loss:
            self.compute_losses()

    def test(self, gen_points=False):
        with torch.no_grad():
            self.forward()
        return self.output

    def compute_losses(self):
        """base_rendering_model.py:533-662 for the colour / zero-one / l2-size / sparse items.  Single process: the
        reference's values.  Under torch.distributed each mean is over the global batch (sums of per-rank gradients are then
        the single-process gradients)."""
        opt, out, W = self.opt, self.output, pdist.world()
        dev = out["coarse_raycolor"].device
        hit = out["ray_mask"][0] > 0
        hidx = out.get("_hit_index")                      # hit-ray indices from the renderer: indexing without a synchronisation
        dense = out.get("_dense_color")                   # fused colour loss (training steps): (dense ray colours, hit flags, number of hit rays)
        self.loss_total = 0
        for i, name in enumerate(opt.color_loss_items):
            if dense is not None:
                # every item from the dense per-ray tensors, no boolean-mask index (each is a device -> host synchronisation): the ray_masked
                # item through the fused pass (it carries the gradient), the others from the filled image (no gradient: see create_network_models)
                from . import ops
                if name == "ray_masked_coarse_raycolor":
                    n = pdist.global_counts(3 * dense[2], device=dev)[0]
                    loss = ops.color_loss_sum_rays(dense[0], self.gt_image[0], dense[1]) / pdist.at_least_one(n)
                else:
                    with torch.no_grad():
                        key = name[len("ray_miss") + 1:] if name.startswith("ray_miss") else (name[len("ray_masked") + 1:] if name.startswith("ray_masked") else name)
                        sq = (out[key][0] - self.gt_image[0]) ** 2
                        if name.startswith("ray_miss"):
                            loss = (sq * torch.logical_not(hit)[:, None]).sum() / 3.0
                        elif name.startswith("ray_masked"):
                            n = pdist.global_counts(3 * dense[2], device=dev)[0]
                            loss = (sq * hit[:, None]).sum() / pdist.at_least_one(n)
                        else:
                            n = pdist.global_counts(sq.numel(), device=dev)[0]
                            loss = sq.sum() / pdist.at_least_one(n)
                self.loss_total = self.loss_total + (loss * opt.color_loss_weights[i] + 1e-6 / W)
                setattr(self, "loss_" + name, loss)
                continue
            if name.startswith("ray_masked"):
                key = name[len("ray_masked") + 1:]
                pred = self._raw[key][0] if (self._raw is not None and key == "coarse_raycolor") else out[key][0][hit]
                gt = self.gt_image[0].index_select(0, hidx) if hidx is not None else self.gt_image[0][hit]
                n = pdist.global_counts(pred.numel(), device=dev)[0]
                loss = ((pred - gt) ** 2).sum() / pdist.at_least_one(n)
            elif name.startswith("ray_miss"):
                key = name[len("ray_miss") + 1:]
                miss = torch.logical_not(hit)
                pred, gt = out[key][0][miss], self.gt_image[0][miss]
                # l2loss(...) * masked_gt.shape[1]  ==  sum of squares / 3   (:559-562)
                loss = ((pred - gt) ** 2).sum() / 3.0          # a SUM over rays: per-rank parts add up over ranks as they are
            else:
                pred, gt = out[name], self.gt_image
                n = pdist.global_counts(pred.numel(), device=dev)[0]
                loss = ((pred - gt) ** 2).sum() / pdist.at_least_one(n)
            self.loss_total = self.loss_total + (loss * opt.color_loss_weights[i] + 1e-6 / W)
            setattr(self, "loss_" + name, loss)
        for i, name in enumerate(opt.zero_one_loss_items):
            if name == "conf_coefficient" and "_zero_one_sum" in out:  # fused into the render node (training steps)
                zsum, count = out["_zero_one_sum"]
                n = pdist.global_counts(count, device=dev)[0]
                loss = zsum / pdist.at_least_one(n)
                self.loss_total = self.loss_total + loss * opt.zero_one_loss_weights[i]
                setattr(self, "loss_" + name, loss)
                continue
            if name == "conf_coefficient" and "_zero_one" in out:      # fused form (NeuralPointsRayMarching.fused_zero_one)
                from . import ops
                conf, pidx_dense, ray_hit, count = out["_zero_one"]
                n = pdist.global_counts(count, device=dev)[0]
                loss = ops.zero_one_conf_sum_rays(conf, pidx_dense, ray_hit, opt.zero_epsilon) / pdist.at_least_one(n)
                self.loss_total = self.loss_total + loss * opt.zero_one_loss_weights[i]
                setattr(self, "loss_" + name, loss)
                continue
            if name not in out:
                continue
            val = torch.clamp(out[name], opt.zero_epsilon, 1 - opt.zero_epsilon)
            n = pdist.global_counts(val.numel(), device=dev)[0]
            loss = (torch.log(val) + torch.log(1 - val)).sum() / pdist.at_least_one(n)
            self.loss_total = self.loss_total + loss * opt.zero_one_loss_weights[i]
            setattr(self, "loss_" + name, loss)
        for i, name in enumerate(opt.l2_size_loss_items):
            n = pdist.global_counts(out[name].numel(), device=dev)[0]
            loss = (out[name] ** 2).sum() / pdist.at_least_one(n)
            self.loss_total = self.loss_total + loss * opt.l2_size_loss_weights[i]
            setattr(self, "loss_" + name, loss)
        if opt.sparse_loss_weight > 0:
            if W > 1:
                raise NotImplementedError("sparse loss is a ratio of two batch sums; not sharded")
            w, cc = out["weight"], out["conf_coefficient"]
            loss = torch.sum(w * torch.abs(1 - torch.exp(-2 * cc))) / (torch.sum(w) + 1e-6)
            out.pop("weight"); out.pop("conf_coefficient")
            self.loss_total = self.loss_total + loss * opt.sparse_loss_weight
            self.loss_sparse = loss

    def backward(self, iters):
        """mvs_points_volumetric_model.py:98-118 (feedforward == 0 branch) + the gradient all-reduce under torch.distributed."""
        for o in self.optimizers:
            o.zero_grad()
        if not self.opt.is_train:
            return
        if isinstance(self.loss_total, torch.Tensor):
            self.loss_total.backward()
        else:
            print("Loss == 0")
            return
        pdist.allreduce_grads(self.net_params, self.neural_params)
        a = self.opt.alter_step
        if (a == 0 or int(iters / a) % 2 == 0) and self.optimizer is not None:
            self.optimizer.step()
        if (a == 0 or int(iters / a) % 2 == 1) and self.neural_point_optimizer is not None:
            self.neural_point_optimizer.step()

    def optimize_parameters(self, backward=True, total_steps=0):
        """neural_points_volumetric_model.py:214-217."""
        self.forward()
        self.update_rank_ray_miss(total_steps)
        self.backward(total_steps)

    # ------------------------------------------------------------------ ray-miss ranking (mvs_points_volumetric_model.py:135-170)
    def update_rank_ray_miss(self, total_steps):
        opt = self.opt
        if getattr(opt, "prob_freq", 0) <= 0 or self.top_ray_miss_loss is None:
            return
        pk = getattr(opt, "prob_kernel_size", None)
        if pk is not None and np.sum(np.asarray(opt.prob_tiers) < total_steps) >= (len(pk) // 3):
            return
        miss = getattr(self, "loss_ray_miss_coarse_raycolor", None)
        if miss is None:
            return
        miss = miss.detach()
        if pdist.world() > 1:
            # the item is a SUM over this rank's rays: every rank must rank the views by the same (global) number, or the ranks pick
            # different probe frames, grow different points and the replicated cloud diverges
            miss = miss.clone()
            torch.distributed.all_reduce(miss)
        if opt.prob_num_step > 1:
            self.top_ray_miss_loss, self.top_ray_miss_ids = self.rank_ray_miss(self.input["id"][0], miss, self.top_ray_miss_ids,
                                                                                  self.top_ray_miss_loss)
        else:
            self.top_ray_miss_loss[0] = torch.maximum(miss, self.top_ray_miss_loss[0])

    def rank_ray_miss(self, new_id, newloss, inds, losses):
        """Keep the ``num_probe`` training views with the largest missed-ray loss, sorted descending; the last slot is the
        scratch entry a new view enters through (:147-156)."""
        with torch.no_grad():
            new_id = int(new_id)
            mask = (inds - new_id) == 0
            if torch.sum(mask) > 0:
                losses[mask] = torch.maximum(newloss.to(losses), losses[mask])
            else:
                inds[-1] = new_id
                losses[-1] = newloss
            losses, order = torch.sort(losses, descending=True)
            return losses, inds[order]

    def reset_ray_miss_ranking(self):
        self.top_ray_miss_loss = torch.zeros([self.num_probe + 1], dtype=torch.float32, device=self.device)
        self.top_ray_miss_ids = torch.arange(self.num_probe + 1, dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------ point-cloud mutators (mvs_points_volumetric_model.py:172-182,240-242)
    def set_points(self, points_xyz, points_embedding, points_color=None, points_dir=None, points_conf=None, Rw2c=None, eulers=None,
                   editing=False):
        if editing:
            raise NotImplementedError("editing_set_points (scene editing) is outside the hot path's callers")
        self.neural_points.set_points(points_xyz, points_embedding, points_color=points_color, points_dir=points_dir,
                                      points_conf=points_conf, parameter=self.opt.feedforward == 0, Rw2c=Rw2c, eulers=eulers)
        if self.opt.feedforward == 0 and self.opt.is_train:
            self.setup_optimizer(self.opt)

    def prune_points(self, thresh):
        self.neural_points.prune(thresh)

    def grow_points(self, points_xyz, points_embedding, points_color, points_dir, points_conf):
        self.neural_points.grow_points(points_xyz, points_embedding, points_color, points_dir, points_conf)

    def gen_points(self, *a, **k):
        raise NotImplementedError("MVSNet point generation is out of scope (SURVEY.md 8f f4)")

    query_embedding = set_bg = gen_points

    # ------------------------------------------------------------------ reporting / checkpoints (base_model.py:66-127)
    def get_current_visuals(self, data=None):
        skip = ("gt_image_ray_masked", "ray_depth_masked_gt_image", "ray_depth_masked_coarse_raycolor", "ray_masked_coarse_raycolor")
        ret = {n: getattr(self, n, None) if n != "gt_image" else self.gt_image for n in self.visual_names if n not in skip}
        if "coarse_raycolor" not in self.visual_names:
            ret["coarse_raycolor"] = self.coarse_raycolor
        return ret

    def get_current_losses(self):
        return {n: getattr(self, "loss_" + n) for n in self.loss_names if hasattr(self, "loss_" + n)}

    def save_networks(self, epoch, other_states={}, back_gpu=True):
        """``<save_dir>/<epoch>_net_ray_marching.pth`` = the network's state_dict on the CPU (keys as in
        tests/test_checkpoint_format.py), ``<epoch>_states.pth`` = the caller's bookkeeping dict.  The network itself stays
        on the device (the reference moves it to the CPU and back)."""
        os.makedirs(self.save_dir, exist_ok=True)
        for name, net in zip(self.model_names, self.get_networks()):
            torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()},
                       os.path.join(self.save_dir, "{}_net_{}.pth".format(epoch, name)))
        torch.save(other_states, os.path.join(self.save_dir, "{}_states.pth".format(epoch)))

    def load_networks(self, epoch):
        """mvs_points_volumetric_model.py:308-326: non-strict load from ``opt.resume_dir``; a "best" checkpoint without stored
        confidences gets ``default_conf``."""
        replaced = False
        for name, net in zip(self.model_names, self.get_networks()):
            path = os.path.join(self.opt.resume_dir, "{}_net_{}.pth".format(epoch, name))
            if not os.path.isfile(path):
                print("cannot load", path)
                continue
            sd = torch.load(path, map_location=self.device)
            dc = getattr(self.opt, "default_conf", -1.0)
            if epoch == "best" and name == "ray_marching" and 0.0 < dc <= 1.0 and self.neural_points.points_conf is not None \
                    and "neural_points.points_conf" not in sd:
                sd["neural_points.points_conf"] = torch.ones_like(self.neural_points.points_conf) * dc
            sd = {k: v for k, v in sd.items() if not (k.startswith("neural_points.") and
                                                      getattr(self.neural_points, k.split(".", 1)[1], None) is None)}
            own = net.state_dict()
            for k, v in sd.items():              # a checkpoint with a different point count replaces the parameters
                if k.startswith("neural_points.") and k in own and own[k].shape != v.shape:
                    attr = k.split(".", 1)[1]
                    old = getattr(self.neural_points, attr)
                    new = nn.Parameter(v.to(self.device))
                    new.requires_grad = old.requires_grad
                    setattr(self.neural_points, attr, new)
                    replaced = True
            net.load_state_dict(sd, strict=False)
        if self.device.type == "cuda":
            self.aggregator.flatten_()
        if self.is_train and self.optimizers and replaced:
            # new parameter objects: the optimizers are rebuilt around them, and the schedulers with them -- a scheduler left on a
            # discarded optimizer decays a learning rate nobody reads (the reference keeps its optimizers across load_state_dict)
            steps = [s.last_epoch for s in getattr(self, "schedulers", [])]
            self.setup_optimizer(self.opt)
            if steps:
                self.init_scheduler(max(steps), self.opt)

    def cleanup(self):
        if getattr(self, "neural_points", None) is not None:
            self.neural_points.querier.clean_up()
        for a in ("neural_points", "net_ray_marching", "aggregator", "output", "input", "gt_image", "_raw"):
            setattr(self, a, None)
        self.clean_optimizer_scheduler()


def create_model(opt):
    """models/__init__.py:36-41 for ``--model mvs_points_volumetric``."""
    if getattr(opt, "model", "mvs_points_volumetric") != "mvs_points_volumetric":
        raise NotImplementedError("model [%s]: only mvs_points_volumetric (every point-nerf script's model) is built" % opt.model)
    m = MvsPointsVolumetricModel()
    m.initialize(opt)
    print("model [{}] was created".format(m.name()))
    return m
