"""Initial per-point appearance from GIVEN 2-D maps: the part of ``models/mvs/mvs_points_model.py`` ``MvsPointsModel``
that feeds the hot path and needs no network -- ``extract_2d`` (:198-218) and ``query_embedding`` (:225-259) -- with the reference's
signatures, argument meaning and return arity, computed by libpnerf_hip.so (csrc/embed2d.hip: projection into the source views, in-image /
z-buffer masks of ``homo_warp_nongrid`` / ``homo_warp_nongrid_occ`` models/mvs/mvs_utils.py:299-315,333-369, bilinear ``grid_sample`` of
``extract_from_2d_grid`` :411-421, the per-view unit directions :239-251).  ``img_feats`` -- the source images and the feature pyramid of the
reference's FeatureNet -- are inputs; the 2-D / MVS networks, ``gen_points`` and the depth filters stay outside the hot-path scope and raise.
Batch size 1, like every reference script (``full_src_feat[0, mask[0,:,0], :]``, mvs_utils.py:419)."""
import ctypes

import torch
import torch.nn as nn

from . import _lib as L, ops

feature_str_lst = ['appr_feature_str0', 'appr_feature_str1', 'appr_feature_str2', 'appr_feature_str3']


def _host(t, n):
    a = t.detach().to("cpu", torch.float32).contiguous().reshape(-1)
    assert a.numel() == n, (tuple(t.shape), n)
    return a.tolist()


def premlp_init(opt):
    """models/mvs/mvs_points_model.py:22-35 (the 63 -> point_features_dim MLP some scripts put behind the sampled features; checkpoint keys
    ``premlp.*``).  Default torch initialisation here; the reference applies its ``init_seq`` on top."""
    in_channels, blocks = 63, []
    act = getattr(nn, opt.act_type, None)
    for _ in range(opt.shading_feature_mlp_layer1):
        blocks += [nn.Linear(in_channels, opt.point_features_dim), act(inplace=True)]
        in_channels = opt.point_features_dim
    return nn.Sequential(*blocks)


class MvsPointsModel(nn.Module):
    """``args`` needs: depth_occ, ref_vid, shading_feature_mlp_layer0 (and layer1 / act_type / point_features_dim when > 0),
    appr_feature_str0..3 (lists such as ["imgfeat_0_0123", "dir_0", "point_conf"])."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        if getattr(args, "shading_feature_mlp_layer0", 0) > 0:
            self.premlp = premlp_init(args)

    # ---- out of scope: the networks -------------------------------------------------------------------------------------------------
    def get_image_features(self, imgs):
        raise NotImplementedError("FeatureNet (models/mvs/models.py) is outside the hot-path scope: pass img_feats in")

    def gen_points(self, batch):
        raise NotImplementedError("MVSNet depth estimation (models/mvs/mvs_points_model.py:262-) is outside the hot-path scope")

    # ---- the path ---------------------------------------------------------------------------------------------------------------------
    def extract_2d(self, img_feats, view_ids, layer_ids, intrinsics, c2ws, w2cs, cam_xyz, HD, WD, cam_vid=0, return_mask=False):
        """-> (out_feats [1,N,sum C of layers != 0 per view], colors [1,N,3 per view] or None); :198-218.  img_feats[lid]: [V,C,H,W]."""
        ops._need_cuda(cam_xyz, "cam_xyz")
        if cam_xyz.shape[0] != 1:
            raise NotImplementedError("extract_2d: batch size 1 (the reference's scatter-back is batch-1 code, mvs_utils.py:419)")
        dev = cam_xyz.device
        xyz = cam_xyz.detach()[0].contiguous().float()
        n, nv = xyz.shape[0], len(view_ids)
        if nv > 8 or nv * len(layer_ids) > 32:
            raise NotImplementedError("extract_2d: at most 8 views / 32 maps per call")
        views = (L.ViewDesc * nv)()
        maps = (L.MapDesc * (nv * len(layer_ids)))()
        keep, fcol, ccol, m = [], 0, 0, 0
        for k, vid in enumerate(view_ids):
            views[k].c2w[:] = _host(c2ws[0, cam_vid], 16)
            views[k].w2c[:] = _host(w2cs[0, vid], 16)
            views[k].intrinsic[:] = _host(intrinsics[0, vid], 9)
            views[k].has_w2c = 0 if vid == cam_vid else 1
            for lid in layer_ids:
                fm = img_feats[lid][vid].detach().contiguous().float()
                if fm.device != dev:
                    raise RuntimeError("pointnerf_amd: img_feats must live on the points' device")
                keep.append(fm)
                C, H, W = fm.shape
                maps[m].d_map, maps[m].view, maps[m].C, maps[m].H, maps[m].W = fm.data_ptr(), k, C, H, W
                maps[m].is_color = 1 if lid == 0 else 0
                maps[m].out_col = ccol if lid == 0 else fcol
                if lid == 0:
                    ccol += C
                else:
                    fcol += C
                m += 1
        occ = 1 if self.args.depth_occ > 0 else 0
        lib = L.lib()
        nws = lib.pnerf_extract_2d_workspace_bytes(n, nv, int(HD), int(WD), occ)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        feats = torch.empty(1, n, fcol, dtype=torch.float32, device=dev)
        colors = torch.empty(1, n, ccol, dtype=torch.float32, device=dev) if ccol else None
        mask = torch.empty(nv, n, dtype=torch.uint8, device=dev) if return_mask else None
        L.check(lib.pnerf_extract_2d(ops._ptr(xyz), n, views, nv, maps, m, int(HD), int(WD), occ, 0.1, ops._ptr(feats), fcol, ops._ptr(colors),
                                     ccol, ops._ptr(mask), ops._ptr(ws), nws, ops._stream()), "pnerf_extract_2d")
        del keep
        if return_mask:
            return feats, colors, mask.bool()
        return feats, colors

    def point_dirs(self, cam_xyz, view_ids, c2ws, w2cs, cam_vid, pointdir_w=False):
        """the "dir" block of query_embedding (:239-251) -> [1,N,3 per view]"""
        ops._need_cuda(cam_xyz, "cam_xyz")
        xyz = cam_xyz.detach()[0].contiguous().float()
        n, nv = xyz.shape[0], len(view_ids)
        ids = torch.as_tensor(list(view_ids), dtype=torch.int64)
        c2w_h, w2c_h = c2ws[0].detach().cpu().float(), w2cs[0].detach().cpu().float()
        pos_cam = (c2w_h[ids, :, 3] @ w2c_h[cam_vid].transpose(0, 1))[..., :3].contiguous()              # :243-245, 4 x 4 host arithmetic
        fa = lambda t, k: (ctypes.c_float * k)(*t.reshape(-1).tolist())
        out = torch.empty(1, n, 3 * nv, dtype=torch.float32, device=xyz.device)
        L.check(L.lib().pnerf_point_dirs(ops._ptr(xyz), n, fa(pos_cam, 3 * nv), nv, fa(c2w_h[cam_vid, :3, :3], 9),
                                         None if pointdir_w else fa(c2w_h[self.args.ref_vid, :3, :3], 9), ops._ptr(out), ops._stream()),
                "pnerf_point_dirs")
        return out

    def query_embedding(self, HDWD, cam_xyz, photometric_confidence, img_feats, c2ws, w2cs, intrinsics, cam_vid, pointdir_w=False):
        """-> (points_embedding, points_colors, points_dirs, points_conf); :225-259"""
        HD, WD = HDWD
        points_embedding, points_dirs, points_conf, points_colors = [], None, None, None
        for feat_str in getattr(self.args, feature_str_lst[cam_vid]):
            if feat_str.startswith("imgfeat"):
                _, view_ids, layer_ids = feat_str.split("_")
                twoD_feats, points_colors = self.extract_2d(img_feats, [int(a) for a in view_ids], [int(a) for a in layer_ids], intrinsics,
                                                            c2ws, w2cs, cam_xyz, HD, WD, cam_vid=cam_vid)
                points_embedding.append(twoD_feats)
            elif feat_str.startswith("dir"):
                points_dirs = self.point_dirs(cam_xyz, [int(a) for a in feat_str.split("_")[1]], c2ws, w2cs, cam_vid, pointdir_w)
            elif feat_str.startswith("point_conf"):
                if photometric_confidence is None:
                    photometric_confidence = torch.ones_like(points_embedding[0][..., 0:1])
                points_conf = photometric_confidence
        points_embedding = torch.cat(points_embedding, dim=-1) if len(points_embedding) > 1 else points_embedding[0]
        if getattr(self.args, "shading_feature_mlp_layer0", 0) > 0:
            points_embedding = self.premlp(torch.cat([points_embedding, points_colors, points_dirs, points_conf], dim=-1))
        return points_embedding, points_colors, points_dirs, points_conf
