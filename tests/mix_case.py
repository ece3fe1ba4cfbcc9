"""The mixed-format tile GEMM of csrc/mixq.h restated in numpy (test infrastructure): f16 h.h over all columns, the two cross terms of the
columns below 256 on e4m3 factors with the formats, scales and block structure of the packed image, the classic three f16 products from
column 256 on.  Shared by the host-emulator test (bit-level check of layouts and scales: the emulated kernel must agree with this to fp32
accumulation noise) and the -m gpu test (the device against float64)."""
import numpy as np


def q_e4m3(v):
    """value -> nearest OCP e4m3fn value (round to nearest even, subnormal step 2^-9, saturating at 448: the device under MODE.FP16_OVFL)"""
    v = np.asarray(v, np.float64)
    a, s = np.abs(v), np.sign(v)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -30)))
    e = np.clip(e, -6, 8)
    step = 2.0 ** (e - 3)
    q = np.round(a / step)                      # numpy rounds half to even
    return s * np.minimum(q * step, 448.0)


def f16(v):
    return np.clip(np.asarray(v, np.float32), -65504, 65504).astype(np.float16).astype(np.float32)


def build(K, seed=11, big=False):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((64, K)) * np.where(rng.random((64, K)) < 0.5, 1.0, 0.01)).astype(np.float32)        # LeakyReLU-like activations
    if big:
        x *= np.float32(40.0)
    w = (rng.uniform(-1, 1, (256, K)) / np.sqrt(K) * 1.7).astype(np.float32)
    x[:, K - 4:] = [1.0, 0.0, 0.0, 0.0]         # a ones column (the bias) and zero padding, like the tiles
    return x, w


def restate(x, w):
    """[64][256] in float64: what the kernel's arithmetic yields with exact accumulation"""
    K = x.shape[1]
    xh = f16(x); xm = f16(x - xh)
    wh = f16(w); wm32 = (w - wh).astype(np.float32)
    out = xh[:, :256].astype(np.float64) @ wh[:, :256].astype(np.float64).T
    x8h = q_e4m3(xh[:, :256]); x8m = q_e4m3(xm[:, :256].astype(np.float64) * 2048.0)
    for blk in range(8):                        # 32 columns = one e4m3 fragment (both lane halves): one block exponent per (row, fragment)
        c = slice(32 * blk, 32 * blk + 32)
        mx = np.maximum(np.abs(wh[:, c]).max(1), np.abs(wm32[:, c]).max(1) * 2048.0).astype(np.float32)
        fr, ex = np.frexp(mx)
        be = np.where(mx > 0, np.where(fr > 0.875, ex - 8, ex - 9), 0).clip(-100, 100).astype(np.float64)
        sc = 2.0 ** be[:, None]
        w8m = q_e4m3(wm32[:, c].astype(np.float64) * 2048.0 / sc); w8h = q_e4m3(wh[:, c].astype(np.float64) / sc)
        out += ((x8h[:, c] @ w8m.T) + (x8m[:, c] @ w8h.T)) * (sc.T / 2048.0)
    if K > 256:
        wm16 = f16(wm32)
        t = slice(256, K)
        out += xh[:, t].astype(np.float64) @ wh[:, t].astype(np.float64).T + xh[:, t].astype(np.float64) @ wm16[:, t].astype(np.float64).T \
            + xm[:, t].astype(np.float64) @ wh[:, t].astype(np.float64).T
    return out


def exact(x, w):
    return x.astype(np.float64) @ w.astype(np.float64).T, np.abs(x.astype(np.float64)) @ np.abs(w.astype(np.float64)).T
