"""dev / what-comes-next evidence: numerical estimate of a 256-term dot product whose two cross terms h.m of the two-plane scheme run
on fp8 (OCP e4m3, block scales of 32 along K) instead of f16 -- against the shipped three f16 products, the two-product inference option and one product.
Pure numpy, no GPU.  Errors are relative to sum |x_k w_k| (what a downstream 1e-4 bar on sigma / RGB sees per layer)."""
import numpy as np

rng = np.random.default_rng(0)


def q_e4m3(v):
    """round to nearest OCP e4m3fn (3 mantissa bits, min normal 2^-6, subnormal step 2^-9, largest 448)"""
    a, s = np.abs(v), np.sign(v)
    e = np.clip(np.floor(np.log2(np.maximum(a, 1e-300))), -6, 8)
    step = 2.0 ** (e - 3)
    return s * np.minimum(np.round(a / step) * step, 448.0)


def blockq(v, blk=32):
    out = np.empty_like(v)
    for r in range(v.shape[0]):
        for b in range(0, v.shape[1], blk):
            seg = v[r, b:b + blk]
            m = np.abs(seg).max()
            sc = 2.0 ** np.ceil(np.log2(m / 448.0)) if m > 0 else 1.0
            out[r, b:b + blk] = q_e4m3(seg / sc) * sc
    return out


def split(v):
    h = v.astype(np.float16).astype(np.float32)
    return h, (v - h).astype(np.float16).astype(np.float32)


K, R = 256, 4096
x = rng.standard_normal((R, K)).astype(np.float32) * np.where(rng.random((R, K)) < 0.5, 1.0, 0.01)     # LeakyReLU-like activations
w = (rng.standard_normal((R, K)) * 0.08).astype(np.float32)
xh, xm = split(x)
wh, wm = split(w)
d = lambda a, b: (a.astype(np.float64) * b).sum(1)
exact, scale = d(x, w), np.abs(x.astype(np.float64) * w).sum(1)
rows = {
    "three f16 products (shipped)": d(xh, wh) + d(xh, wm) + d(xm, wh),
    "f16 h.h + fp8 h.m + fp8 m.h (block scales of 32)": d(xh, wh) + d(blockq(xh), blockq(wm)) + d(blockq(xm), blockq(wh)),
    "two f16 products (weights' residual dropped: the inference option)": d(xh, wh) + d(xm, wh),
    "one f16 product": d(xh, wh),
}
for name, val in rows.items():
    e = (val - exact) / scale
    print("%-70s rms %.2e  max %.2e" % (name, np.sqrt(np.mean(e ** 2)), np.abs(e).max()))
