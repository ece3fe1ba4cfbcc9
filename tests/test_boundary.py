"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly the symbols
include/pnerf.h declares; the Python binding lists the same set; the product never imports the oracle and fails
loudly (no fallback) when the library or a GPU is missing."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "pnerf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(pnerf_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    from pointnerf_amd import _lib
    lib_path = _lib.LIB_PATH
    if not os.path.exists(lib_path):
        import __graft_entry__ as g
        g.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path]).decode()
    exported = set(re.findall(r"\bT (pnerf_[a-z0-9_]+)", out))
    declared = _header_symbols()
    assert declared, "no declarations parsed from include/pnerf.h"
    assert declared <= exported, "declared but not exported: %s" % sorted(declared - exported)
    assert exported <= declared, "exported but not declared in the header: %s" % sorted(exported - declared)
    assert set(_lib.PROTOTYPES) == declared, sorted(set(_lib.PROTOTYPES) ^ declared)


def test_library_loads_and_reports_gfx950():
    from pointnerf_amd import _lib
    lib = _lib.lib()
    assert lib.pnerf_arch() == b"gfx950" and lib.pnerf_version() >= 1000
    # pure host-side queries work without a GPU
    import ctypes
    offs = (ctypes.c_int64 * (_lib.MLP_NTENSORS + 1))()
    assert lib.pnerf_mlp_layout(32, offs) == 0 and offs[_lib.MLP_NTENSORS] == 341764
    assert lib.pnerf_mlp_layout(16, offs) == -4            # unsupported feature width is an error, not a fallback
    gp = _lib.GridParams()
    gp.vdim[:] = [162, 290, 189]
    assert lib.pnerf_grid_workspace_bytes(ctypes.byref(gp), 2_000_000) > 162 * 290 * 189 * 4
    # the saved-activation area: ~5.4 KB per neighbor row + ~4.4 KB per sample (one f16 plane per saved GEMM operand: DESIGN.md 4.1)
    assert 100000 * 8 * 5400 < lib.pnerf_agg_saved_bytes(100000, 8) < 100000 * 8 * 6200


def test_code_object_is_gfx950_only():
    from pointnerf_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_80", b"sm_90"):
        assert other not in blob


def test_code_object_has_no_packed_fp32_math(tmp_path):
    """csrc/Makefile builds without the vectorisers: on the MI355X a v_pk_fma_f32 can deliver a wrong low destination register for the
    wave's last 16 lanes while another wave of the SIMD executes MFMA (isolated by tools/pkfma_probe.hip, profiles/r04_pkfma_probe.log;
    tests/test_gpu_pkfma_probe.py runs the probe).  The shipped code object must not contain packed fp32 arithmetic at all -- whatever
    compiler or flag change would bring it back fails here."""
    import struct
    from pointnerf_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no ROCm llvm tools here")
    blob = open(_lib.LIB_PATH, "rb").read()
    # the device code objects are ELF files (e_machine 224 = AMDGPU) embedded in the host library, one per translation unit
    pos, n_obj, mfma, packed = 1, 0, 0, []
    while True:
        pos = blob.find(b"\x7fELF", pos)
        if pos < 0:
            break
        if struct.unpack_from("<H", blob, pos + 18)[0] == 224:
            shoff, = struct.unpack_from("<Q", blob, pos + 40)
            shentsize, shnum = struct.unpack_from("<HH", blob, pos + 58)
            co = tmp_path / ("co%d.elf" % n_obj)
            co.write_bytes(blob[pos: pos + shoff + shentsize * shnum])
            asm = subprocess.check_output([objdump, "-d", "--mcpu=gfx950", str(co)]).decode()
            mfma += asm.count("v_mfma_f32_32x32x16_f16")
            packed += re.findall(r"v_pk_(?:fma|mul|add)_f32", asm)
            n_obj += 1
        pos += 4
    assert n_obj >= 8 and mfma > 1000, (n_obj, mfma)                  # it IS the kernels' code
    assert not packed, len(packed)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pointnerf_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt and "/root/reference" not in txt, f


def test_missing_library_is_a_hard_error(monkeypatch):
    from pointnerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpnerf_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.lib()


def test_ops_refuse_cpu_tensors():
    from pointnerf_amd import ops, config
    from pointnerf_amd.point_aggregators import PointAggregator
    gp = ops.make_grid_params([0] * 6, [1, 1, 1], [2, 2, 2], [3, 3, 3], [3, 3, 3], 9, 100, 0.1)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        ops.build_grid(gp, torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        PointAggregator(config.lego_opt()).flatten_()
    with pytest.raises(NotImplementedError):
        PointAggregator(config.lego_opt(agg_distance_kernel="quadric"))


def test_ctypes_structs_mirror_the_header(tmp_path):
    """The PODs cross the C ABI by pointer: the ctypes mirrors must have the header's sizes and field offsets (plain C, gcc)."""
    import ctypes
    from pointnerf_amd import _lib
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pnerf.h"\nint main(void) {\n'
                   'printf("%zu %zu %zu %zu\\n", sizeof(pnerf_grid_params), sizeof(pnerf_camera), sizeof(pnerf_points), sizeof(pnerf_point_grads));\n'
                   'printf("%zu %zu %zu\\n", offsetof(pnerf_point_grads, ready_event), offsetof(pnerf_points, n), offsetof(pnerf_camera, has_bg));\n'
                   'return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes, offs = [list(map(int, l.split())) for l in subprocess.check_output([str(exe)]).decode().splitlines()]
    assert sizes == [ctypes.sizeof(_lib.GridParams), ctypes.sizeof(_lib.Camera), ctypes.sizeof(_lib.Points), ctypes.sizeof(_lib.PointGrads)]
    assert offs == [_lib.PointGrads.ready_event.offset, _lib.Points.n.offset, _lib.Camera.has_bg.offset]


def test_querier_reads_query_size_at_call_time(monkeypatch):
    """Pins the one deliberate behavioural difference of the querier that changes probe results: the reference caches
    ``opt.query_size`` at construction (point_query.py:39-42) and so ignores ``probe_hole``'s enlargement of it (run/train_ft.py:428);
    this querier reads it per call, i.e. the probe really runs with ``prob_kernel_size``."""
    from pointnerf_amd import ops, config
    from pointnerf_amd.point_query import lighting_fast_querier, clear_grid_cache
    seen = []
    monkeypatch.setattr(ops, "grid_hyperparameters", lambda opt, xyz: ([0] * 6, [1, 1, 1], [4, 4, 4], 0.1))
    monkeypatch.setattr(ops, "make_grid_params", lambda rg, vs, vd, ks, qs, P, max_o, radius: seen.append(tuple(int(q) for q in qs)) or object())
    monkeypatch.setattr(ops, "build_grid", lambda gp, xyz: object())
    opt = config.lego_opt()
    q = lighting_fast_querier(torch.device("cpu"), opt)
    xyz = torch.zeros(1, 10, 3)
    clear_grid_cache()
    q._grid(xyz, 10)
    opt.query_size = [5, 5, 5]                     # what probe_hole does between two queries of the same cloud
    q._grid(xyz, 10)
    assert seen == [(3, 3, 3), (5, 5, 5)], seen   # a new grid, dilated with the new size: not the cached one, not the constructor's value
    clear_grid_cache()


def test_point_embedding_has_no_cpu_path():
    import types
    from pointnerf_amd.mvs_points_model import MvsPointsModel
    m = MvsPointsModel(types.SimpleNamespace(depth_occ=0, ref_vid=0, shading_feature_mlp_layer0=0))
    z = torch.zeros(1, 4, 3)
    with pytest.raises(RuntimeError, match="device tensor"):
        m.extract_2d([torch.zeros(1, 3, 8, 8)], [0], [0], torch.eye(3)[None, None], torch.eye(4)[None, None], torch.eye(4)[None, None], z, 8, 8)
    with pytest.raises(NotImplementedError):
        m.gen_points({})
