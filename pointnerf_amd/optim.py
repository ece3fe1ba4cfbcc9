"""Parameter update of the hot path's caller: the reference steps two ``torch.optim.Adam`` instances per iteration
(models/mvs_points_volumetric_model.py:80-91 build them, :98-118 step them: MLP parameters with ``opt.lr``, the
``neural_points.*`` parameters with ``opt.plr``, betas (0.9, 0.999)).  ``FusedAdam`` is that optimizer on
libpnerf_hip.so (pnerf_adam_step: one pass over p, g, m, v per tensor) with torch's state layout
(``state[p] = {step, exp_avg, exp_avg_sq}``), so ``state_dict()`` / ``load_state_dict()`` interchange with
``torch.optim.Adam`` checkpoints.  SURVEY.md 8(f2).

``ShardedAdam`` is the ZeRO-1 form for ray-sharded data parallelism: gradients are reduce-scattered (RCCL over xGMI), every
rank updates 1/G of each flattened parameter and the updated shards are all-gathered -- the same bytes on the wire as
all-reducing the gradients, 1/G of the Adam work and state per GPU."""
import ctypes

import torch

from . import _lib as L


def _adam_hip(p, g, m, v, lr, b1, b2, eps, step):
    lib = L.lib()
    n = p.numel()
    for t in (p, g, m, v):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.numel() == n):
            raise ValueError("FusedAdam: parameters, gradients and state must be contiguous fp32 device tensors of one size")
    L.check(lib.pnerf_adam_step(ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(m.data_ptr()),
                                ctypes.c_void_p(v.data_ptr()), n, lr, b1, b2, eps, step,
                                ctypes.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)), "pnerf_adam_step")


def _adam_hip_multi(items, b1, b2, eps):
    """items: [(p, g, m, v, lr, step)] -> ONE pnerf_adam_step_multi call (one launch per 24 tensors)"""
    lib = L.lib()
    arr = (L.AdamTensor * len(items))()
    for a, (p, g, m, v, lr, step) in zip(arr, items):
        n = p.numel()
        for t in (p, g, m, v):
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.numel() == n):
                raise ValueError("FusedAdam: parameters, gradients and state must be contiguous fp32 device tensors of one size")
        a.param, a.grad, a.exp_avg, a.exp_avg_sq, a.n, a.lr, a.step = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, step
    L.check(lib.pnerf_adam_step_multi(arr, len(items), b1, b2, eps, ctypes.c_void_p(torch.cuda.current_stream(items[0][0].device).cuda_stream)),
            "pnerf_adam_step_multi")


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) semantics (no weight decay, no amsgrad): one HIP pass over p, g, m, v, and ONE launch for
    all tensors that share (betas, eps) -- every group of this optimizer in the training loop's use."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, update=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._update = update                       # tests on CPU inject a per-tensor torch restatement; the product path is the HIP kernel

    @torch.no_grad()
    def collect(self):
        """Advance the step counters and return the update list {(b1, b2, eps): [(p, g, m, v, lr, step)]} WITHOUT applying it
        (``step`` / ``step_all`` launch it)."""
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                batches.setdefault((float(b1), float(b2), float(group["eps"])), []).append(
                    (p.data, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), int(st["step"].item())))
        return batches

    def _apply(self, batches):
        for (b1, b2, eps), items in batches.items():
            if self._update is not None:
                for (p, g, m, v, lr, step) in items:
                    self._update(p, g, m, v, lr, b1, b2, eps, step)
            else:
                _adam_hip_multi(items, b1, b2, eps)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._apply(self.collect())
        return loss


def step_all(optimizers):
    """Step several optimizers; the FusedAdam instances among them with ONE launch per (betas, eps) -- the reference's two optimizers
    (MLP at lr, points at plr, same betas and eps) become one kernel.  Equivalent to calling ``step()`` on each."""
    merged, host = {}, None
    for o in optimizers:
        if isinstance(o, FusedAdam) and o._update is None:
            for k, items in o.collect().items():
                merged.setdefault(k, []).extend(items)
            host = o
        else:
            o.step()
    if host is not None:
        host._apply(merged)


class ShardedAdam:
    """ZeRO-1 Adam over a process group: state and update are sharded by rank, parameters stay replicated.

    The parameters are re-homed at construction into ONE persistent flat buffer (every ``p.data`` becomes a view of it; the buffer
    is padded to world x shard floats), and a second persistent flat buffer of the same layout receives the gradients.  A step
    is then: copy the gradients into their slots (skipped for a gradient that already IS its slot: see ``grad_views``), ONE
    reduce-scatter straight out of the gradient buffer, the Adam update of this rank's shard in place in the parameter buffer,
    ONE all-gather straight into the parameter buffer -- no per-step allocation, no padded copies of parameters or gradients.
    Shard boundaries cross tensor boundaries (Adam is element-wise), so the shards are equal whatever the tensor sizes."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, group=None, update=None):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.group = lr, betas, eps, group
        self._update = update or _adam_hip
        self.step_count = 0
        dist = torch.distributed
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        dev = self.params[0].device
        self.offsets, o = [], 0
        for p in self.params:
            self.offsets.append(o)
            o += (p.numel() + 3) // 4 * 4                 # every tensor starts on a 16-byte boundary
        self.shard = ((o + self.world - 1) // self.world + 3) // 4 * 4
        total = self.shard * self.world
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.exp_avg = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.shard, dtype=torch.float32, device=dev)

    def grad_views(self):
        """The gradient slots, one per parameter: a producer that writes its gradients there (``p.grad = view``) saves the copy."""
        return [self.flat_grad[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)]

    @torch.no_grad()
    def step(self):
        """One sharded update.  Two contracts differ from torch.optim.Adam / FusedAdam and are checked or stated here:
          * a parameter whose ``.grad`` is None contributes a ZERO gradient (the collective needs every slot on every rank) and is still
            updated -- its moments decay and it keeps moving on momentum -- whereas Adam skips it.  Freeze a tensor by taking it out of
            the optimizer (rebuild it: the training loop does that on prune / grow anyway), not by leaving its gradient unset;
          * the parameters must still live in ``flat_param``: anything that replaces ``p.data`` / the Parameter objects (prune, grow,
            ``flatten_()``) detaches them from this optimizer -- rebuild it afterwards.  Checked below."""
        dist = torch.distributed
        self.step_count += 1
        base = self.flat_param.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != base + 4 * o:
                raise RuntimeError("ShardedAdam: a parameter no longer lives in the optimizer's flat buffer (its storage was replaced after "
                                   "construction, e.g. by prune / grow / flatten_()): rebuild the optimizer")
            slot = self.flat_grad[o:o + p.numel()]
            if p.grad is None:
                slot.zero_()
            elif p.grad.data_ptr() != slot.data_ptr():
                slot.copy_(p.grad.reshape(-1))
        W, shard = self.world, self.shard
        mine = slice(self.rank * shard, (self.rank + 1) * shard)
        gs = self.flat_grad[mine]
        from . import dist as pdist
        comm = W > 1 or (pdist.FORCE_COLLECTIVES and dist.is_initialized())      # (one rank + the bring-up switch: the collectives run as identities)
        if comm:
            if dist.get_backend(self.group) == "gloo":       # gloo has no reduce_scatter: all-reduce in place (CPU tests only)
                dist.all_reduce(self.flat_grad, group=self.group)
            else:
                dist.reduce_scatter_tensor(gs, self.flat_grad, group=self.group)      # in place: the output is this rank's slice of the input
        ps = self.flat_param[mine]
        self._update(ps, gs, self.exp_avg, self.exp_avg_sq, float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                     self.step_count)
        if comm:
            if dist.get_backend(self.group) == "gloo":
                dist.all_gather(list(self.flat_param.view(W, shard).unbind(0)), ps.clone(), group=self.group)
            else:
                dist.all_gather_into_tensor(self.flat_param, ps, group=self.group)     # in place: the input is this rank's slice of the output

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
