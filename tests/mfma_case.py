"""One 32x32x16 f16 MFMA with operands that exercise what csrc/f16x3.h relies on: the fragment layout and f16 SUBNORMAL
inputs (the residual plane of any activation below ~0.06 is subnormal: if the matrix pipe flushed them the two-plane
product would lose 2^-14 instead of 2^-25 per term).  Shared by the emulated (CPU) and the -m gpu test."""
import numpy as np


def build():
    rng = np.random.default_rng(7)
    A = rng.standard_normal((32, 16)).astype(np.float16)
    B = rng.standard_normal((16, 32)).astype(np.float16)
    # half of A's entries subnormal (|x| < 6.1e-5), paired with large B so that they matter in the result
    sub = (rng.integers(1, 1000, (32, 16)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float16)
    mask = rng.random((32, 16)) < 0.5
    A = np.where(mask, sub, A * np.float16(1e-4))
    B = (B * np.float16(512.0)).astype(np.float16)
    a = np.zeros((64, 8), np.float16); b = np.zeros((64, 8), np.float16)
    for l in range(64):
        a[l] = A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8]
        b[l] = B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31]
    D = A.astype(np.float64) @ B.astype(np.float64)
    # what a pipe that flushes subnormal INPUTS would return
    Af = np.where(np.abs(A.astype(np.float32)) < 2.0 ** -14, 0, A).astype(np.float64)
    return a, b, D, Af @ B.astype(np.float64)


def unpack(d):
    out = np.zeros((32, 32), np.float64)
    for l in range(64):
        for r in range(16):
            out[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = d[l, r]
    return out


def split_inputs():
    """values for the two-plane split: wide dynamic range, subnormal-range values, the f16 boundary, out-of-range values"""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(65536) * 10.0 ** rng.uniform(-9, 4.5, 65536)).astype(np.float32)
    x[:8] = [0.0, -0.0, 65504.0, -65504.0, 65519.9, 1e-8, -3e-8, 6.1e-5]
    return x


def split_expected(x, sat):
    from test_split_f16_cpu import f16_rtz
    x = np.asarray(x, np.float32)
    if sat:         # the gradient form (pn_split2_sat): clamp, then the high plane rounded to nearest even (v_cvt_pk_f16_f32)
        x = np.clip(x, -65504.0, 65504.0)
        h = x.astype(np.float16)
    else:
        h = f16_rtz(x)
    with np.errstate(over="ignore"):
        m = (x - h.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h, m
