"""pnerf_adam_step (one-pass HIP Adam) against torch.optim.Adam on the device: same update to fp32 rounding,
tensor sizes that exercise the float4 body and the scalar tail, state interchange with a torch checkpoint."""
import pytest
import torch

from pointnerf_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shapes", [[(8192, 32), (8192, 1), (8192, 3)], [(1000003,)], [(7,), (256, 284), (1,)]])
def test_fused_adam_matches_torch_adam(shapes):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    a = [torch.randn(s, generator=gen).to(dev).requires_grad_(True) for s in shapes]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    oa, ob = torch.optim.Adam(a, lr=2e-3, betas=(0.9, 0.999)), FusedAdam(b, lr=2e-3, betas=(0.9, 0.999))
    for it in range(6):
        for p, q in zip(a, b):
            g = torch.randn(p.shape, generator=gen).to(dev) * (10.0 ** (it - 3))       # gradients over six decades
            p.grad, q.grad = g, g.clone()
        oa.step(); ob.step()
        for p, q in zip(a, b):
            err = float((p - q).abs().max())
            assert err <= 2e-6, (it, tuple(p.shape), err)
    sa = oa.state_dict()
    for k, st in ob.state_dict()["state"].items():
        assert float(st["step"]) == 6.0
        for name in ("exp_avg", "exp_avg_sq"):                 # moments: to fp32 rounding of the tensor's scale
            ref = sa["state"][k][name]
            assert float((st[name] - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), name


def test_fused_adam_on_views_of_a_flat_vector():
    """PointAggregator.flatten_() makes every parameter a view into one flat vector: the views are only 4-byte aligned."""
    dev = torch.device("cuda:0")
    flat_a = torch.randn(1000, device=dev); flat_b = flat_a.clone()
    a = [flat_a[1:258].view(257).requires_grad_(True), flat_a[258:999].view(741).requires_grad_(True)]
    b = [flat_b[1:258].view(257).requires_grad_(True), flat_b[258:999].view(741).requires_grad_(True)]
    oa, ob = torch.optim.Adam(a, lr=1e-2), FusedAdam(b, lr=1e-2)
    for it in range(3):
        for p, q in zip(a, b):
            g = torch.randn(p.shape, device=dev)
            p.grad, q.grad = g, g.clone()
        oa.step(); ob.step()
    assert float((flat_a - flat_b).abs().max()) <= 2e-6


def test_fused_adam_rejects_what_the_kernel_cannot_take():
    p = torch.zeros(16, dtype=torch.float64, device="cuda:0", requires_grad=True)
    p.grad = torch.ones_like(p)
    with pytest.raises(ValueError):
        FusedAdam([p]).step()


def test_step_all_is_one_launch_and_equals_separate_steps():
    """the reference's two optimizers (18 MLP tensors at lr, 4 point tensors at plr) stepped by ONE pnerf_adam_step_multi launch
    (plus the 24-tensor split): same parameters as torch.optim.Adam on each, more than 24 tensors and an empty one included"""
    from pointnerf_amd import ops
    from pointnerf_amd.optim import step_all
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1)
    shapes_a = [(256, 284), (256,), (256, 256), (256,), (256, 263), (256,), (1, 256), (1,), (128, 280), (128,), (3, 128), (3,)] + [(5,)] * 15 + [(0,)]
    shapes_b = [(1, 50000, 32), (1, 50000, 1), (1, 50000, 3), (1, 50001, 3)]
    mk = lambda shapes: [torch.randn(s, generator=gen).to(dev).requires_grad_(True) for s in shapes]
    a1, b1 = mk(shapes_a), mk(shapes_b)
    a2, b2 = [p.detach().clone().requires_grad_(True) for p in a1], [p.detach().clone().requires_grad_(True) for p in b1]
    ta, tb = torch.optim.Adam(a1, lr=5e-4), torch.optim.Adam(b1, lr=2e-3)
    fa, fb = FusedAdam(a2, lr=5e-4), FusedAdam(b2, lr=2e-3)
    for it in range(4):
        for p, q in zip(a1 + b1, a2 + b2):
            g = torch.randn(p.shape, generator=gen).to(dev)
            p.grad, q.grad = g, g.clone()
        if it == 2:
            a1[3].grad = a2[3].grad = None                 # a parameter without a gradient is skipped (like torch.optim.Adam)
        ta.step(); tb.step()
        ops.prof_enable(True); ops.prof_collect()
        step_all([fa, fb])
        launches = ops.prof_collect()["adam"][1]
        ops.prof_enable(False)
        assert launches == 1, launches                      # one profiler scope = one pnerf_adam_step_multi call
        for p, q in zip(a1 + b1, a2 + b2):
            assert float((p - q).abs().max()) <= 2e-6 if p.numel() else True
    assert float(fa.state[a2[3]]["step"]) == 3.0 and float(fa.state[a2[0]]["step"]) == 4.0
