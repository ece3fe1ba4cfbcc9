"""oracle/query.py -- ctypes bindings of the CPU query checkers.  TEST INFRASTRUCTURE.

  oracle_query : oracle/liboracle_query.so  (our C restatement, oracle/query_oracle.c)
  ref_query    : oracle/_ref/libref_query.so (the reference's own kernels run serially on the host,
                 oracle/ref_driver.cpp; only K <= 8, the reference's hard-coded KN)
Both implement the native op woord_query_grid_point_index
(/root/reference/models/neural_points/cuda/query_worldcoords.cpp:34-82) on numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def build(force=False):
    """Compile the checkers (gcc/g++ only).  _ref is rebuilt only where /root/reference exists."""
    so = os.path.join(_HERE, "liboracle_query.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "query_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_query.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libref_query.so")
    if os.path.isdir("/root/reference") and (force or not os.path.exists(ref_so)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _lib(name):
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _libs[name] = ctypes.CDLL(path)
    return _libs[name]


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_query.so")) or os.path.isdir("/root/reference")


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _common(raypos, xyz, kernel_size, query_size, vdim, ranges, vsize):
    raypos = np.ascontiguousarray(raypos, dtype=np.float32)
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    ks = np.ascontiguousarray(kernel_size, dtype=np.int32)
    qs = np.ascontiguousarray(query_size, dtype=np.int32)
    vd = np.ascontiguousarray(vdim, dtype=np.int32)
    rg = np.ascontiguousarray(ranges, dtype=np.float32)
    vs = np.ascontiguousarray(vsize, dtype=np.float32)
    return raypos, xyz, ks, qs, vd, rg, vs


def oracle_query(raypos, xyz, kernel_size, query_size, SR, K, vdim, max_o, P, radius, ranges, vsize,
                 n_actual=None, nthreads=1):
    """raypos [R,D,3], xyz [N,3] -> (sample_pidx [R2,SR,K] i32, sample_loc [R2,SR,3] f32, ray_mask [R] i8, info)."""
    lib = _lib("liboracle_query.so")
    raypos, xyz, ks, qs, vd, rg, vs = _common(raypos, xyz, kernel_size, query_size, vdim, ranges, vsize)
    R, D = raypos.shape[:2]
    N = xyz.shape[0]
    pidx = np.empty((max(R, 1), SR, K), np.int32)
    loc = np.empty((max(R, 1), SR, 3), np.float32)
    mask = np.zeros((max(R, 1),), np.int8)
    info = np.zeros(8, np.int32)
    r2 = ctypes.c_int(0)
    f, i = ctypes.c_float, ctypes.c_int
    rc = lib.pnerf_oracle_query(_p(raypos, f), _p(xyz, f), i(N), i(N if n_actual is None else n_actual),
                                _p(ks, i), _p(qs, i), i(SR), i(K), i(R), i(D), _p(vd, i), i(max_o), i(P),
                                f(float(radius)), _p(rg, f), _p(vs, f), i(nthreads), _p(pidx, i), _p(loc, f),
                                _p(mask, ctypes.c_byte), ctypes.byref(r2), _p(info, i))
    if rc != 0:
        raise RuntimeError("pnerf_oracle_query failed: %d" % rc)
    names = ["n_occ", "max_cnt", "ovf_max_o", "ovf_P", "R1", "R2", "n_sel", "n_neigh"]
    return pidx[:r2.value].copy(), loc[:r2.value].copy(), mask[:R].copy(), dict(zip(names, info.tolist()))


def ref_query(raypos, xyz, kernel_size, query_size, SR, K, vdim, max_o, P, radius, ranges, vsize,
              n_actual=None, nthreads=1, T=1024):
    """Same contract, computed by the reference's own kernels (serial host build)."""
    lib = _lib(os.path.join("_ref", "libref_query.so"))
    raypos, xyz, ks, qs, vd, rg, vs = _common(raypos, xyz, kernel_size, query_size, vdim, ranges, vsize)
    R, D = raypos.shape[:2]
    N = xyz.shape[0]
    pidx = np.empty((max(R, 1), SR, K), np.int32)
    loc = np.empty((max(R, 1), SR, 3), np.float32)
    mask = np.zeros((max(R, 1),), np.int8)
    r2, hits = ctypes.c_int(0), ctypes.c_int(0)
    f, i = ctypes.c_float, ctypes.c_int
    rc = lib.pnerf_ref_query(_p(raypos, f), _p(xyz, f), i(N), i(N if n_actual is None else n_actual),
                             _p(ks, i), _p(qs, i), i(SR), i(K), i(R), i(D), _p(vd, i), i(max_o), i(P),
                             f(float(radius)), _p(rg, f), _p(vs, f), i(T), _p(pidx, i), _p(loc, f),
                             _p(mask, ctypes.c_byte), ctypes.byref(r2), ctypes.byref(hits))
    if rc != 0:
        raise RuntimeError("pnerf_ref_query failed: %d" % rc)
    return pidx[:r2.value].copy(), loc[:r2.value].copy(), mask[:R].copy(), dict(curand_hits=hits.value, R2=r2.value)


def bruteforce(xyz, c, radius, cap=4096):
    lib = _lib("liboracle_query.so")
    xyz = np.ascontiguousarray(xyz, np.float32); c = np.ascontiguousarray(c, np.float32)
    idx = np.empty(cap, np.int32); d2 = np.empty(cap, np.float32)
    n = lib.pnerf_oracle_bruteforce(_p(xyz, ctypes.c_float), ctypes.c_int(xyz.shape[0]), _p(c, ctypes.c_float),
                                    ctypes.c_float(float(radius)), ctypes.c_int(cap), _p(idx, ctypes.c_int),
                                    _p(d2, ctypes.c_float))
    return idx[:min(n, cap)].copy(), d2[:min(n, cap)].copy(), n
