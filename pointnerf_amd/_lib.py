"""ctypes binding of libpnerf_hip.so (C ABI in include/pnerf.h).

The library is the product: there is NO CPU or PyTorch fallback.  If the shared object is missing
or a symbol does not resolve, importing an op raises immediately.  ``import torch`` happens first
on purpose: libpnerf_hip.so needs ``libamdhip64.so.7`` and must bind to the one PyTorch-ROCm has
already loaded, so that device pointers and streams of torch's allocator are valid in our launches.
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch's HIP runtime before ours resolves libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpnerf_hip.so")

c_int, c_i64, c_f32, c_void_p, c_size_t = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

PNERF_MAX_K = 16
GI_N_IN_GRID, GI_N_OCC, GI_MAX_CNT, GI_CELL0, GI_FIRST_IDX, GI_LEN = 0, 1, 2, 3, 4, 8
MLP_NTENSORS = 18

ERRORS = {-1: "PNERF_E_INVAL (bad argument)", -2: "PNERF_E_WS (workspace too small)",
          -3: "PNERF_E_LAUNCH (HIP launch/runtime error)", -4: "PNERF_E_UNSUP (unsupported configuration)"}


class GridParams(ctypes.Structure):
    _fields_ = [("ranges", c_f32 * 6), ("vsize", c_f32 * 3), ("vdim", ctypes.c_int32 * 3),
                ("kernel_size", ctypes.c_int32 * 3), ("query_size", ctypes.c_int32 * 3),
                ("P", ctypes.c_int32), ("max_o", ctypes.c_int32), ("radius", c_f32)]


class Camera(ctypes.Structure):
    _fields_ = [("campos", c_f32 * 3), ("camrot", c_f32 * 9), ("rw2c", c_f32 * 9), ("vsize_z", c_f32),
                ("raydist_mode_unit", ctypes.c_int32), ("bg", c_f32 * 3), ("has_bg", ctypes.c_int32)]


class Points(ctypes.Structure):
    _fields_ = [("xyz", c_void_p), ("embedding", c_void_p), ("conf", c_void_p), ("dir", c_void_p),
                ("color", c_void_p), ("n", ctypes.c_int32), ("feat_dim", ctypes.c_int32)]


class AdamTensor(ctypes.Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("n", ctypes.c_int64), ("lr", ctypes.c_double), ("step", ctypes.c_int64)]


class ViewDesc(ctypes.Structure):
    _fields_ = [("c2w", c_f32 * 16), ("w2c", c_f32 * 16), ("intrinsic", c_f32 * 9), ("has_w2c", ctypes.c_int32)]


class MapDesc(ctypes.Structure):
    _fields_ = [("d_map", c_void_p), ("view", ctypes.c_int32), ("C", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("out_col", ctypes.c_int32), ("is_color", ctypes.c_int32), ("first_of_view", ctypes.c_int32)]


class PointGrads(ctypes.Structure):
    _fields_ = [("embedding", c_void_p), ("conf", c_void_p), ("dir", c_void_p), ("color", c_void_p), ("ready_event", c_void_p),
                ("zero_one_gscale", c_void_p), ("zero_one_eps", c_f32)]


# symbol -> (restype, argtypes); kept in lock-step with include/pnerf.h (tests/test_boundary.py checks it)
PROTOTYPES = {
    "pnerf_version": (c_int, []),
    "pnerf_arch": (ctypes.c_char_p, []),
    "pnerf_grid_workspace_bytes": (c_size_t, [ctypes.POINTER(GridParams), c_int]),
    "pnerf_grid_build": (c_int, [ctypes.POINTER(GridParams), c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "pnerf_grid_info": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int32), c_void_p]),
    "pnerf_points_minmax": (c_int, [c_void_p, c_i64, c_void_p, c_void_p]),
    "pnerf_query_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pnerf_query": (c_int, [ctypes.POINTER(GridParams), c_void_p, c_void_p, ctypes.POINTER(c_f32), c_void_p, c_void_p,
                            c_f32, c_f32, c_f32, ctypes.c_uint64, c_int, c_int, c_int, c_int,
                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pnerf_gather_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_i64, c_void_p, c_void_p]),
    "pnerf_scatter_add_rows": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_int, c_void_p]),
    "pnerf_zero_one_blocks": (c_int, [c_i64]),
    "pnerf_zero_one_forward": (c_int, [c_void_p, c_int, c_void_p, c_i64, c_f32, c_void_p, c_void_p]),
    "pnerf_zero_one_backward": (c_int, [c_void_p, c_int, c_void_p, c_i64, c_f32, c_void_p, c_void_p, c_void_p]),
    "pnerf_color_loss_blocks": (c_int, [c_int]),
    "pnerf_color_loss_forward_rays": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "pnerf_color_loss_backward_rays": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "pnerf_zero_one_forward_rays": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_f32, c_void_p, c_void_p]),
    "pnerf_zero_one_backward_rays": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_f32, c_void_p, c_void_p, c_void_p]),
    "pnerf_voxel_downsample_workspace_bytes": (c_size_t, [c_i64, c_int, c_int, c_int]),
    "pnerf_voxel_downsample": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pnerf_extract_2d_workspace_bytes": (c_size_t, [c_i64, c_int, c_int, c_int, c_int]),
    "pnerf_extract_2d": (c_int, [c_void_p, c_i64, ctypes.POINTER(ViewDesc), c_int, ctypes.POINTER(MapDesc), c_int, c_int, c_int, c_int, c_f32,
                                 c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pnerf_point_dirs": (c_int, [c_void_p, c_i64, ctypes.POINTER(c_f32), c_int, ctypes.POINTER(c_f32), ctypes.POINTER(c_f32), c_void_p, c_void_p]),
    "pnerf_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_i64, c_void_p]),
    "pnerf_adam_step_multi": (c_int, [ctypes.POINTER(AdamTensor), c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_void_p]),
    "pnerf_mlp_layout": (c_int, [c_int, ctypes.POINTER(c_i64)]),
    "pnerf_mlp_packed_bytes": (c_size_t, []),
    "pnerf_mlp_pack": (c_int, [c_void_p, c_void_p, c_void_p]),
    "pnerf_agg_saved_bytes": (c_size_t, [c_i64, c_int]),
    "pnerf_agg_workspace_bytes": (c_size_t, [c_i64, c_int]),
    "pnerf_render_backward_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pnerf_render_forward": (c_int, [ctypes.POINTER(Camera), ctypes.POINTER(Points), c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_i64, c_void_p, c_size_t, c_void_p]),
    "pnerf_set_inference_products": (c_int, [c_int]),
    "pnerf_set_wgrad_planes": (c_int, [c_int]),
    "pnerf_set_cross_terms": (c_int, [c_int]),
    "pnerf_set_cross_terms_where": (c_int, [c_int]),
    "pnerf_compact_workspace_bytes": (c_size_t, [c_i64]),
    "pnerf_touched_flags": (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "pnerf_compact_valid": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pnerf_agg_forward": (c_int, [ctypes.POINTER(Camera), ctypes.POINTER(Points), c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_size_t, c_void_p]),
    "pnerf_agg_backward": (c_int, [ctypes.POINTER(Camera), ctypes.POINTER(Points), c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_i64,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(PointGrads),
                                   c_void_p, c_size_t, c_void_p]),
    "pnerf_raymarch_forward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_f32), c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pnerf_raymarch_backward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_f32), c_int, c_int,
                                        c_void_p, c_void_p, c_void_p]),
    "pnerf_debug_mfma_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "pnerf_debug_mix_gemm": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pnerf_debug_mfma_rate": (c_int, [c_int, c_int, c_void_p, ctypes.POINTER(ctypes.c_double), c_void_p]),
    "pnerf_debug_uniform": (c_int, [ctypes.c_uint64, ctypes.c_uint64, c_i64, c_void_p, c_void_p]),
    "pnerf_debug_split": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_int, c_void_p]),
    "pnerf_debug_pe": (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "pnerf_prof_enable": (c_int, [c_int]),
    "pnerf_prof_kernel_count": (c_int, []),
    "pnerf_prof_kernel_name": (ctypes.c_char_p, [c_int]),
    "pnerf_prof_collect": (c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
    "pnerf_render_backward": (c_int, [ctypes.POINTER(Camera), ctypes.POINTER(Points), c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_int, c_int, c_i64,
                                      c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, ctypes.POINTER(PointGrads),
                                      c_void_p, c_size_t, c_void_p]),
}

_lib = None


def lib():
    """The loaded library; raises (never falls back) if it is absent or incomplete."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "pointnerf_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for this path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise RuntimeError("pointnerf_amd: libpnerf_hip.so lacks symbol %s (stale build?)" % name) from e
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("pointnerf_amd: %s failed: %s" % (what, ERRORS.get(rc, rc)))
