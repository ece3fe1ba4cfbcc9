"""Run the Python host layer on top of the HOST-EMULATED kernels (tools/emu): the same C ABI as libpnerf_hip.so, built from
the same .hip sources for x86, every GPU thread a fiber.  Test infrastructure only (like oracle/): nothing in
pointnerf_amd/ knows about it; the `emu_backend` context manager patches the ctypes handle and the device checks of
pointnerf_amd.ops for the duration of a test, so that CPU tensors flow through the unmodified host code into the emulated
kernels.  Used by tests/test_emu_*.py (-m "not gpu"): kernel index / layout / synchronisation bugs show up without a GPU."""
import contextlib
import ctypes
import importlib.util
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_handle = None


def emu_lib():
    global _handle
    if _handle is None:
        spec = importlib.util.spec_from_file_location("build_emu", os.path.join(ROOT, "tools", "emu", "build_emu.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        path = mod.build()
        from pointnerf_amd import _lib as L
        h = ctypes.CDLL(path)
        for name, (res, args) in L.PROTOTYPES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _handle = h
    return _handle


@contextlib.contextmanager
def emu_backend(ncu=2):
    from pointnerf_amd import _lib as L, ops
    import gpu_util
    h = emu_lib()
    saved = (L._lib, ops._ptr, ops._stream, ops._need_cuda, gpu_util.DEV, os.environ.get("PN_EMU_NCU"))

    def _ptr(t):
        if t is None:
            return ctypes.c_void_p(0)
        assert t.is_contiguous()
        return ctypes.c_void_p(t.data_ptr())

    L._lib, ops._ptr, ops._stream, ops._need_cuda, gpu_util.DEV = h, _ptr, (lambda: ctypes.c_void_p(0)), (lambda t, n: None), "cpu"
    os.environ["PN_EMU_NCU"] = str(ncu)
    try:
        yield h
    finally:
        L._lib, ops._ptr, ops._stream, ops._need_cuda, gpu_util.DEV = saved[:5]
        if saved[5] is None:
            os.environ.pop("PN_EMU_NCU", None)
        else:
            os.environ["PN_EMU_NCU"] = saved[5]
