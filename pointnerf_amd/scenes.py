"""Seeded synthetic scenes and ray batches.

No dataset or checkpoint is reachable from the build or GPU box, so the parity tests and
``bench.py`` use generators that emit exactly the tensors the reference's dataset + checkpoint
would hand to the hot path (Appendix B of SURVEY.md: ``campos``, ``camrotc2w``, ``raydir``,
``pixel_idx``, ``near``, ``far``, ``intrinsic``, ``h``, ``w``, ``bg_color``; point cloud
``xyz/points_embeding/points_conf/points_dir/points_color``).  Camera maths restates
``data/load_blender.py:29-59`` (pose_spherical, blender2opencv) and
``data/data_utils.py:55-70`` (get_dtu_raydir with dir_norm=False).
"""
import numpy as np

_B2O = np.diag([1.0, -1.0, -1.0, 1.0])


def pose_spherical(theta_deg, phi_deg, radius):
    """c2w (4x4, float64) in the blender convention, data/load_blender.py:51-56."""
    th, ph = theta_deg / 180.0 * np.pi, phi_deg / 180.0 * np.pi
    trans = np.eye(4, dtype=np.float32); trans[2, 3] = radius
    rphi = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0],
                     [0, 0, 0, 1]], dtype=np.float32)
    rth = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                    [0, 0, 0, 1]], dtype=np.float32)
    swap = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    return swap @ (rth @ (rphi @ trans))


def synth_camera(theta_deg, phi_deg=-30.0, radius=4.0, w=800, h=800, camera_angle_x=0.6911112070083618):
    """NeRF-synthetic pinhole (nerf_synth360_ft_dataset.py:378-395): returns c2w(4x4 f32), K(3x3 f32)."""
    focal = 0.5 * 800 / np.tan(0.5 * camera_angle_x) * (w / 800.0)
    c2w = (pose_spherical(theta_deg, phi_deg, radius) @ _B2O).astype(np.float32)
    intr = np.array([[focal, 0, w / 2], [0, focal, h / 2], [0, 0, 1]], dtype=np.float32)
    return c2w, intr


def rays_for_pixels(c2w, intr, px, py):
    """get_dtu_raydir(dir_norm=False): un-normalised world ray dirs, shape [R,3] float32."""
    x = (px.astype(np.float32) + 0.5 - intr[0, 2]) / intr[0, 0]
    y = (py.astype(np.float32) + 0.5 - intr[1, 2]) / intr[1, 1]
    dirs = np.stack([x, y, np.ones_like(x)], axis=-1).astype(np.float32)
    return (dirs @ c2w[:3, :3].T).astype(np.float32)


def ray_dict(c2w, intr, px, py, near=2.0, far=6.0, w=800, h=800, gt_seed=0):
    """The per-step input dict of Appendix B (numpy, batch dim 1 added)."""
    R = px.size
    raydir = rays_for_pixels(c2w, intr, px.reshape(-1), py.reshape(-1))
    gt = np.random.default_rng(gt_seed).random((R, 3), dtype=np.float32)
    return dict(
        campos=c2w[:3, 3][None].astype(np.float32),
        camrotc2w=c2w[:3, :3][None].astype(np.float32),
        raydir=raydir[None],
        pixel_idx=np.stack([px.reshape(-1), py.reshape(-1)], -1)[None].astype(np.float32),
        gt_image=gt[None],
        near=np.full((1, 1, 1), near, np.float32), far=np.full((1, 1, 1), far, np.float32),
        intrinsic=intr[None], h=np.array([h]), w=np.array([w]),
        bg_color=np.ones((1, 3), np.float32),
    )


def block_rays(theta_deg=30.0, x0=368, y0=368, size=64, **kw):
    """configs[0]: the size x size pixel block starting at (x0, y0) of pose (theta, -30, 4)."""
    c2w, intr = synth_camera(theta_deg)
    py, px = np.meshgrid(np.arange(y0, y0 + size), np.arange(x0, x0 + size), indexing="ij")
    return ray_dict(c2w, intr, px, py, **kw)


def random_rays(pose_i, R, w=800, h=800):
    """configs[1]: R uniformly random pixels (seed 2+i) of train-like pose i (theta = 3.6 deg * i)."""
    c2w, intr = synth_camera(3.6 * pose_i)
    rng = np.random.default_rng(2 + pose_i)
    px = rng.integers(0, w, size=R); py = rng.integers(0, h, size=R)
    return ray_dict(c2w, intr, px, py, gt_seed=2 + pose_i)


def point_attributes(n, feat_dim=32, seed=0):
    """Per-point learnables with the reference's init ranges (neural_points.py:291; conf in (0.1,1))."""
    rng = np.random.default_rng(1000 + seed)
    emb = (rng.random((1, n, feat_dim), dtype=np.float32) - 0.5)
    conf = (0.1 + 0.9 * rng.random((1, n, 1), dtype=np.float32)).astype(np.float32)
    color = rng.random((1, n, 3), dtype=np.float32)
    d = rng.standard_normal((1, n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return dict(points_embeding=emb, points_conf=conf, points_color=color, points_dir=d.astype(np.float32))


def chair_points(n=8192, seed=0, radius=0.10):
    """configs[0]: n points on a jittered sphere shell of radius 0.10 about the origin."""
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    r = radius + rng.uniform(-0.002, 0.002, size=(n, 1))
    return (d * r).astype(np.float32)


def lego_points(n=2_000_000, seed=1, pitch=0.0075,
                ranges=(-0.638, -1.141, -0.346, 0.634, 1.149, 1.141),
                centre=(0.0, 0.0, 0.40), axes=(0.55, 1.00, 0.65)):
    """configs[1]: n points of a jittered lattice (pitch = the reference's vox_res=320 down-sampling
    pitch) restricted to the thinnest band around an ellipsoid that holds n lattice nodes; lattice
    (x-major) order, i.e. spatially sorted like the reference's voxel down-sampler output
    (models/mvs/mvs_utils.py:537-561)."""
    lo, hi = np.array(ranges[:3]), np.array(ranges[3:])
    ax = [np.arange(lo[a] + 0.5 * pitch, hi[a], pitch) for a in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    P = np.stack([X.ravel(), Y.ravel(), Z.ravel()], -1)
    e = np.sqrt((((P - np.array(centre)) / np.array(axes)) ** 2).sum(-1))
    dev = np.abs(e - 1.0)
    n = min(n, dev.size)
    thr = np.partition(dev, n - 1)[n - 1]
    keep = np.flatnonzero(dev <= thr)[:n]            # lattice order preserved
    rng = np.random.default_rng(seed)
    pts = P[keep] + rng.uniform(-0.5, 0.5, size=(keep.size, 3)) * pitch
    pts = np.clip(pts, lo + 1e-4, hi - 1e-4)
    return pts.astype(np.float32)


# ------------------------------------------------------------------ configs[3] / configs[4]
def _face_slab(rng, lo, hi, axis, side, pitch, layers, keep):
    """Jittered lattice on one axis-aligned face of the box [lo,hi], `layers` lattice layers thick towards the inside."""
    a, b = [i for i in range(3) if i != axis]
    ua = np.arange(lo[a] + 0.5 * pitch, hi[a], pitch)
    ub = np.arange(lo[b] + 0.5 * pitch, hi[b], pitch)
    A, B = np.meshgrid(ua, ub, indexing="ij")
    out = []
    for l in range(layers):
        m = rng.random(A.shape) < keep
        n = int(m.sum())
        p = np.empty((n, 3))
        p[:, a], p[:, b] = A[m], B[m]
        p[:, axis] = (lo[axis] + (l + 0.5) * pitch) if side == 0 else (hi[axis] - (l + 0.5) * pitch)
        out.append(p + rng.uniform(-0.5, 0.5, size=p.shape) * pitch)
    return np.concatenate(out, 0)


def _box_shell(rng, lo, hi, pitch, layers, keep):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    return np.concatenate([_face_slab(rng, lo, hi, ax, sd, pitch, layers, keep) for ax in range(3) for sd in (0, 1)], 0)


def scannet_points(n=6_000_000, seed=3, pitch=0.008):
    """configs[3]: ScanNet-scale room: inner faces of an 8.0 x 6.0 x 2.8 m box (3 lattice layers thick, as noisy COLMAP
    surfaces are) plus 12 axis-aligned cuboids, 30 % random dropout, truncated to n points (face-major order)."""
    rng = np.random.default_rng(seed)
    parts = [_box_shell(rng, (-4.0, -3.0, 0.0), (4.0, 3.0, 2.8), pitch, 3, 0.7)]
    for i in range(12):
        c = np.array([rng.uniform(-3.2, 3.2), rng.uniform(-2.3, 2.3), 0.0])
        sz = np.array([rng.uniform(0.5, 1.6), rng.uniform(0.5, 1.6), rng.uniform(0.4, 1.8)])
        parts.append(_box_shell(rng, c - [sz[0] / 2, sz[1] / 2, 0], c + [sz[0] / 2, sz[1] / 2, sz[2]], pitch, 1, 0.7))
    pts = np.concatenate(parts, 0)
    if pts.shape[0] < n:        # densify with extra wall layers until the count is reached
        extra = _box_shell(rng, (-3.97, -2.97, 0.03), (3.97, 2.97, 2.77), pitch, 6, 0.7)
        pts = np.concatenate([pts, extra], 0)
    return pts[:n].astype(np.float32)


def scannet_rays(pose_i, R, w=640, h=480, f=577.87):
    """configs[3] camera: 50 poses on a 1.2 m circle at height 1.4 m looking outward (OpenCV convention: +z forward, +y down)."""
    th = 2 * np.pi * (pose_i % 50) / 50.0
    fwd = np.array([np.cos(th), np.sin(th), 0.0]); down = np.array([0.0, 0.0, -1.0]); right = np.cross(down, fwd)
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2] = right, down, fwd
    c2w[:3, 3] = [1.2 * np.cos(th), 1.2 * np.sin(th), 1.4]
    intr = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], dtype=np.float32)
    rng = np.random.default_rng(300 + pose_i)
    px = rng.integers(0, w, size=R); py = rng.integers(0, h, size=R)
    return ray_dict(c2w, intr, px, py, near=0.1, far=8.0, w=w, h=h, gt_seed=300 + pose_i)


BARN_RANGES = (-2.05965, -0.48064, -2.23660, 1.78036, 0.6094, 1.28341)


def barn_points(n=20_000_000, seed=4, pitch=0.003):
    """configs[4]: Tanks&Temples-Barn-scale shell: the faces of a box inset in the Barn `ranges`, thick enough (whole
    lattice layers) to hold n points, truncated to n (face-major order)."""
    rng = np.random.default_rng(seed)
    lo = np.array(BARN_RANGES[:3]) + 0.05; hi = np.array(BARN_RANGES[3:]) - 0.05
    ext = hi - lo
    area = 2 * (ext[0] * ext[1] + ext[0] * ext[2] + ext[1] * ext[2])
    layers = int(np.ceil(n / (area / pitch ** 2))) + 1
    pts = _box_shell(rng, lo, hi, pitch, layers, 1.0)
    return pts[:n].astype(np.float32)


def barn_rays(pose_i, R, w=1088, h=640):
    """configs[4] camera: 60 poses on an ellipse outside the shell, looking at its centre; focal 0.7 W."""
    th = 2 * np.pi * (pose_i % 60) / 60.0
    ctr = np.array([(BARN_RANGES[0] + BARN_RANGES[3]) / 2, (BARN_RANGES[1] + BARN_RANGES[4]) / 2, (BARN_RANGES[2] + BARN_RANGES[5]) / 2])
    eye = ctr + np.array([3.2 * np.cos(th), -0.3, 3.0 * np.sin(th)])
    fwd = ctr - eye; fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, -1.0, 0.0]); right = np.cross(fwd, up); right /= np.linalg.norm(right); down = np.cross(fwd, right)
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, eye
    intr = np.array([[0.7 * w, 0, w / 2], [0, 0.7 * w, h / 2], [0, 0, 1]], dtype=np.float32)
    rng = np.random.default_rng(400 + pose_i)
    px = rng.integers(0, w, size=R); py = rng.integers(0, h, size=R)
    return ray_dict(c2w, intr, px, py, near=0.01, far=4.5, w=w, h=h, gt_seed=400 + pose_i)
