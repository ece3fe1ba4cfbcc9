"""CPU tests of the query oracle (oracle/query_oracle.c):
  * bit-exact against the reference's OWN kernels compiled for the host and run in canonical serial
    order (oracle/_ref) on randomised scenes incl. ragged / empty / edge cases;
  * brute-force properties that do not depend on any restatement.
"""
import numpy as np
import pytest
import torch

from pointnerf_amd import config, scenes
from oracle import pyref, query as oq

needs_ref = pytest.mark.skipif(not oq.have_ref(), reason="oracle/_ref not built and no /root/reference")


def _scene(seed, n, radius=0.06, size=10, **ov):
    kw = dict(K=8, SR=16, P=12, max_o=50000, ranges=[-0.3, -0.3, -0.3, 0.3, 0.3, 0.3])
    kw.update(ov)
    opt = config.lego_opt(**kw)
    xyz = torch.from_numpy(scenes.chair_points(n, seed=seed, radius=radius))
    inp = pyref.to_torch_inputs(scenes.block_rays(theta_deg=17.0 * seed, x0=400 - size // 2, y0=400 - size // 2, size=size))
    return opt, xyz, inp


def _same(a, b):
    assert a["sample_pidx"].shape == b["sample_pidx"].shape
    assert torch.equal(a["sample_pidx"], b["sample_pidx"])
    assert torch.equal(a["sample_loc_w"], b["sample_loc_w"])       # bit-exact floats
    assert torch.equal(a["ray_mask"], b["ray_mask"])


@needs_ref
@pytest.mark.parametrize("seed,n,K,SR,P", [(0, 1200, 8, 16, 12), (1, 700, 4, 8, 12), (2, 3000, 8, 32, 20),
                                           (3, 300, 1, 4, 12), (4, 2000, 6, 24, 16)])
def test_oracle_equals_reference_kernels(seed, n, K, SR, P):
    opt, xyz, inp = _scene(seed, n, K=K, SR=SR, P=P)
    a = pyref.query(opt, xyz, inp, impl="oracle")
    b = pyref.query(opt, xyz, inp, impl="ref")
    assert b["info"]["curand_hits"] == 0
    assert a["info"]["ovf_P"] == 0 and a["info"]["ovf_max_o"] == 0
    _same(a, b)
    assert a["sample_pidx"].shape[1] > 0


@needs_ref
def test_config1_chair_equals_reference_kernels():
    """BASELINE.json configs[0]: chair 64x64, 8k points, K=4, SR=32."""
    opt = config.chair_opt()
    xyz = torch.from_numpy(scenes.chair_points())
    inp = pyref.to_torch_inputs(scenes.block_rays())
    a = pyref.query(opt, xyz, inp, impl="oracle", nthreads=4)
    b = pyref.query(opt, xyz, inp, impl="ref")
    _same(a, b)
    assert a["sample_pidx"].shape == (1, 3282, 32, 4)


@needs_ref
def test_edge_cases_match_reference():
    # rays that miss everything (camera looks at an empty region)
    opt, xyz, inp = _scene(5, 500)
    inp["raydir"] = inp["raydir"] * torch.tensor([1.0, 1.0, -1.0])
    a, b = pyref.query(opt, xyz, inp, impl="oracle"), pyref.query(opt, xyz, inp, impl="ref")
    _same(a, b)
    assert a["sample_pidx"].shape[1] == 0 and int(a["ray_mask"].sum()) == 0
    # a single point; points outside opt.ranges; radius clipping at the grid border
    opt, xyz, inp = _scene(6, 400, ranges=[-0.05, -0.05, -0.05, 0.05, 0.05, 0.05])
    _same(pyref.query(opt, xyz, inp, impl="oracle"), pyref.query(opt, xyz, inp, impl="ref"))
    opt, xyz, inp = _scene(7, 1)
    _same(pyref.query(opt, xyz, inp, impl="oracle"), pyref.query(opt, xyz, inp, impl="ref"))
    # kernel_size 5 (two layers beyond the centre) and query_size 1 (no dilation)
    opt, xyz, inp = _scene(8, 1500, kernel_size=[5, 5, 5], query_size=[1, 1, 1])
    _same(pyref.query(opt, xyz, inp, impl="oracle"), pyref.query(opt, xyz, inp, impl="ref"))
    # radius_limit_scale = 0 disables the radius test (.cu:272)
    opt, xyz, inp = _scene(9, 800, radius_limit_scale=0)
    _same(pyref.query(opt, xyz, inp, impl="oracle"), pyref.query(opt, xyz, inp, impl="ref"))


def test_voxel0_quirk_is_reproduced():
    """query_worldcoords.cu:147 `voxel_idx > 0`: the cell of the first in-range point holds no points."""
    opt, xyz, inp = _scene(0, 1200)
    q = pyref.query(opt, xyz, inp)
    hp = q["hp"]
    cell = lambda p: np.floor((p - hp["ranges"][:3]) / hp["scaled_vsize"]).astype(np.int64)
    cells = cell(xyz.numpy())
    in0 = np.all(cells == cells[0], axis=1)
    hit = torch.unique(q["sample_pidx"][q["sample_pidx"] >= 0]).numpy()
    assert not np.isin(np.flatnonzero(in0), hit).any()


def test_bruteforce_properties():
    """Every returned neighbor is within the radius; when fewer than K are returned from the centre
    layer the result is a subset of the brute-force ball; -1 padding is a suffix."""
    opt, xyz, inp = _scene(2, 3000, SR=16)
    q = pyref.query(opt, xyz, inp)
    pidx, loc = q["sample_pidx"][0].numpy(), q["sample_loc_w"][0].numpy()
    r = q["hp"]["radius"]
    xn = xyz.numpy()
    rng = np.random.default_rng(0)
    rays = rng.choice(pidx.shape[0], size=min(40, pidx.shape[0]), replace=False)
    checked = 0
    for ri in rays:
        for s in range(pidx.shape[1]):
            ids = pidx[ri, s]
            v = ids[ids >= 0]
            if v.size == 0:
                continue
            assert np.all(ids[v.size:] == -1)                 # -1 only as a suffix
            assert len(set(v.tolist())) == v.size             # no duplicates
            d = xn[v] - loc[ri, s]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            assert np.all(d2.astype(np.float32) <= np.float32(r) * np.float32(r))
            bidx, _, n = oq.bruteforce(xn, loc[ri, s], r)
            assert np.isin(v, bidx).all()
            checked += 1
    assert checked > 50


def test_threads_do_not_change_results():
    opt, xyz, inp = _scene(4, 2000, SR=24)
    _same(pyref.query(opt, xyz, inp, nthreads=1), pyref.query(opt, xyz, inp, nthreads=4))


def test_overflow_is_reported():
    opt, xyz, inp = _scene(1, 4000, radius=0.02, P=2)
    q = pyref.query(opt, xyz, inp)
    assert q["info"]["ovf_P"] == 1 and q["info"]["max_cnt"] > 2
