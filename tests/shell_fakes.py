"""Stand-ins for the objects the reference's run scripts hand to ``test()`` / ``probe_hole()`` (run/train_ft.py:252-530):
a model whose ``test()`` returns seeded, hand-made outputs per ray, a dataset of per-view dicts, and a visualizer that only
records.  Used twice with the SAME seeds: by tests/golden/make_golden.py, which runs the REFERENCE's functions (exec'ed from
their source text) on them, and by the tests, which run pointnerf_amd.probe / eval_loop and the oracle's restatements on
them and compare with the stored results."""
import numpy as np
import torch

from pointnerf_amd import config

PROBE_KEYS = ("ray_max_sample_loc_w", "ray_max_far_dist", "ray_max_shading_opacity", "shading_avg_color", "shading_avg_dir",
              "shading_avg_conf", "shading_avg_embedding", "coarse_raycolor")


class FakeModel:
    """``set_input(data)`` / ``test()`` / ``get_current_visuals(data=)`` / ``output`` / ``visual_names`` / ``opt`` as the run
    scripts use them.  The per-view maps depend on (seed, view id) only, so every caller sees the same "renders"."""

    def __init__(self, opt, H, W, seed, hit_frac=0.6, near_gt=None):
        self.opt, self.device, self.H, self.W, self.seed, self.hit_frac = opt, torch.device("cpu"), H, W, seed, hit_frac
        self.near_gt = near_gt                 # {view id: gt [H*W,3]}: coarse_raycolor = gt + small noise (the far_thresh rule compares them)
        self.visual_names = ["coarse_raycolor", "gt_image", "ray_masked_coarse_raycolor"]
        self.calls, self.seen, self.output = 0, [], None
        self.top_ray_miss_ids = None
        self._maps = {}

    def maps(self, vid):
        if vid not in self._maps:
            g = torch.Generator().manual_seed(1000 * self.seed + int(vid))
            H, W = self.H, self.W
            hit = torch.rand(H, W, generator=g) < self.hit_frac
            hit[0, :] = False; hit[:, 0] = False                                  # misses on the border exercise the clamp of bloat_inds
            m = dict(ray_max_sample_loc_w=torch.rand(H, W, 3, generator=g), ray_max_far_dist=torch.rand(H, W, 1, generator=g) * 0.02,
                     ray_max_shading_opacity=torch.rand(H, W, 1, generator=g), shading_avg_color=torch.rand(H, W, 3, generator=g),
                     shading_avg_dir=torch.rand(H, W, 3, generator=g), shading_avg_conf=torch.rand(H, W, 1, generator=g),
                     shading_avg_embedding=torch.rand(H, W, 32, generator=g), coarse_raycolor=torch.rand(H, W, 3, generator=g))
            if self.near_gt is not None:
                m["coarse_raycolor"] = self.near_gt[int(vid)].reshape(H, W, 3) + 0.08 * (torch.rand(H, W, 3, generator=g) - 0.5)
            self._maps[vid] = (hit, m)
        return self._maps[vid]

    def eval(self):
        return self

    def train(self):
        return self

    def set_input(self, d):
        self.input = d

    def test(self):
        self.calls += 1
        self.seen.append((getattr(self.opt, "prob", 0), tuple(np.asarray(self.opt.query_size).tolist())))
        vid = int(self.input["id"])
        hit_map, maps = self.maps(vid)
        p = self.input["pixel_idx"][0].long()
        hit = hit_map[p[:, 1], p[:, 0]]
        out = {k: (v[p[:, 1], p[:, 0]] * hit[:, None])[None] for k, v in maps.items()}
        bg = self.input["bg_color"].reshape(1, 1, 3)
        out["coarse_raycolor"] = torch.where(hit[None, :, None], out["coarse_raycolor"], bg.expand(1, hit.numel(), 3))   # (fill_invalid)
        out["ray_mask"] = hit[None].to(torch.int8)
        if not bool(hit.any()) or getattr(self.opt, "prob", 0) != 1:
            out = dict(coarse_raycolor=out["coarse_raycolor"], ray_mask=out["ray_mask"])
        self.output = out
        return dict(out)

    def get_current_visuals(self, data=None):
        return {"coarse_raycolor": self.output["coarse_raycolor"], "gt_image": data["gt_image"] if data is not None else None}

    def reset_ray_miss_ranking(self):
        pass


class FakeDataset:
    """``get_item(i)`` / ``len()`` / ``total`` / ``height`` / ``width`` of the reference's datasets in ``no_crop`` form; some
    views provide rays for a sub-rectangle only (``edge_mask`` then matters)."""

    def __init__(self, H, W, n_views, seed, partial=()):
        self.height, self.width, self.total = H, W, n_views
        g = torch.Generator().manual_seed(seed)
        self.views = []
        for i in range(n_views):
            h0, w0 = (2, 3) if i in partial else (0, 0)
            py, px = torch.meshgrid(torch.arange(h0, H), torch.arange(w0, W), indexing="ij")
            pix = torch.stack([px, py], -1)[None].float()                                    # [1,h,w,2] (px, py), row-major
            n = pix.shape[1] * pix.shape[2]
            gt = torch.rand(1, n, 3, generator=g)
            gt[0, torch.rand(n, generator=g) < 0.3] = 1.0                                     # background-coloured pixels are not holes
            self.views.append(dict(raydir=torch.zeros(1, n, 3), pixel_idx=pix, gt_image=gt, bg_color=torch.ones(1, 3), id=i))

    def __len__(self):
        return self.total

    def get_item(self, i):
        return dict(self.views[i])

    def __getitem__(self, i):
        return dict(self.views[i])

    def gt_canvas(self, i):
        """[H*W,3] ground truth scattered by pixel (0 where the view has no ray)"""
        v = self.views[i]
        p = v["pixel_idx"].reshape(-1, 2).long()
        c = torch.zeros(self.height, self.width, 3)
        c[p[:, 1], p[:, 0]] = v["gt_image"][0]
        return c.reshape(-1, 3)


class RecordingVisualizer:
    image_dir = "/nonexistent"

    def __init__(self):
        self.acc, self.shown, self.details = [], [], []

    def reset(self):
        pass

    def print_details(self, s):
        self.details.append(s)

    def display_current_results(self, visuals, i, opt=None):
        self.shown.append((i, {k: np.array(v) for k, v in visuals.items()}))

    def accumulate_losses(self, d):
        self.acc.append({k: float(v) for k, v in d.items()})

    def print_losses(self, count):
        pass

    def get_psnr(self, key):
        import math
        return float(np.mean([-10.0 * math.log(a[key]) / math.log(10.0) for a in self.acc]))

    def save_ref_views(self, *a, **k):
        pass

    def save_neural_points(self, *a, **k):
        pass

    def gen_video(self, *a, **k):
        pass

def vox_cloud(n, seed):
    gen = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, generator=gen) * torch.tensor([1.0, 0.6, 0.3]) + torch.tensor([-0.2, 0.1, 2.0])


def shell_probe_setup(far_thresh):
    """the options / stand-in model / dataset of the probe_hole fixture (tests rebuild them from the same seeds)"""
    H, W = 23, 31
    o = config.lego_train_opt(prob_kernel_size=[5, 5, 5, 7, 7, 7], prob_tiers=[100, 200], prob_mul=0.5, prob_num_step=1, prob_top=1,
                              prob_mode=0, far_thresh=far_thresh, random_sample_size=10, bgmodel="No")
    o.query_size = [3, 3, 3]
    data = FakeDataset(H, W, 4, seed=2, partial=(1,))
    model = FakeModel(o, H, W, seed=1, near_gt={i: data.gt_canvas(i) for i in range(4)} if far_thresh > 0 else None)
    model.top_ray_miss_loss = torch.tensor([0.9, 0.5, 0.0, 0.2, 0.0])
    model.top_ray_miss_ids = torch.tensor([2, 1, 3, 0, 0])
    return o, model, data


def shell_test_setup():
    H, W = 17, 21
    o = config.lego_train_opt(random_sample_size=9, bgmodel="No", test_num_step=2)
    o.test_color_loss_items = ["coarse_raycolor", "ray_masked_coarse_raycolor"]
    o.visual_items = ["coarse_raycolor", "gt_image"]
    o.query_size = [3, 3, 3]
    data = FakeDataset(H, W, 5, seed=6, partial=(2,))
    model = FakeModel(o, H, W, seed=3)
    model.visual_names = ["coarse_raycolor", "gt_image"]
    return o, model, data


# ---- the extract_2d / query_embedding fixture (tests/golden/refembed.npz; tests/golden/make_golden.py --embed) ----
def embed_inputs(seed=5, n=1000, HD=24, WD=32, n_views=3, focal=26.0):
    """seeded inputs of the extract_2d / query_embedding fixture (tests rebuild them with this function): points in the frame of camera
    0 (some outside its image, a few behind it), three cameras on an arc looking at the cloud, an image + 3-level feature pyramid per view"""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(n, 3, generator=g) * torch.tensor([2.4, 1.8, 1.5]) + torch.tensor([-1.2, -0.9, 1.5])
    xyz[: n // 50, 2] = -xyz[: n // 50, 2]                                            # behind the camera
    xyz[n // 50: n // 25] = xyz[n // 25: n // 25 + (n // 25 - n // 50)] * torch.tensor([1.0, 1.0, 1.35])   # occluded twins on the same pixels
    c2ws, w2cs, intr = [], [], []
    for v in range(n_views):
        ang = 0.22 * v
        R = torch.tensor([[np.cos(ang), 0.0, np.sin(ang)], [0.0, 1.0, 0.0], [-np.sin(ang), 0.0, np.cos(ang)]], dtype=torch.float32)
        c2w = torch.eye(4)
        c2w[:3, :3] = R
        c2w[:3, 3] = torch.tensor([0.35 * v, 0.05 * v, -0.1 * v])
        c2ws.append(c2w); w2cs.append(torch.linalg.inv(c2w))
        f = focal + v
        intr.append(torch.tensor([[f, 0.0, (WD - 1) / 2.0 + 0.3 * v], [0.0, f, (HD - 1) / 2.0], [0.0, 0.0, 1.0]]))
    feats = [torch.rand(n_views, c, HD // d, WD // d, generator=g) for c, d in ((3, 1), (8, 1), (16, 2), (32, 4))]
    conf = torch.rand(1, n, 1, generator=g)
    return dict(cam_xyz=xyz[None].contiguous(), c2ws=torch.stack(c2ws)[None], w2cs=torch.stack(w2cs)[None], intrinsics=torch.stack(intr)[None],
                img_feats=feats, photometric_confidence=conf, HD=HD, WD=WD)


EMBED_CASES = {   # tag -> (depth_occ, cam_vid, feature strings of that camera, pointdir_w, pass the confidences)
    "a": (0, 0, ["imgfeat_0_0123", "dir_0", "point_conf"], False, True),
    "b": (1, 0, ["imgfeat_0_0123", "dir_0", "point_conf"], False, False),
    "c": (0, 1, ["imgfeat_012_0123", "dir_012", "point_conf"], True, True),
    "d": (1, 2, ["imgfeat_201_123", "dir_20"], False, True),
}
