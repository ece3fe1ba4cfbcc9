"""CPU check of the arithmetic behind the split-bf16 weight-gradient kernel (pointnerf_amd/csrc/backward.hip: k_wgrad_b3):
an fp32 number is the EXACT sum of three bf16 numbers obtained by round-to-nearest-even conversions of the successive
residuals, and the six products the kernel keeps (ah*bl, ah*bm, ah*bh, am*bm, am*bh, al*bh) reproduce the fp32 product to
2^-23 -- the level of the fp32 accumulation's own rounding.  numpy restatement of b3_split4 (v_cvt_pk_bf16_f32 = RNE)."""
import numpy as np


def bf16_rne(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    l = bf16_rne((r - m).astype(np.float32))
    return h, m, l


def _is_bf16(v):
    return bool(np.all((v.view(np.uint32) & np.uint32(0xFFFF)) == 0))


def test_three_bf16_planes_sum_to_the_fp32_value_exactly():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, 200000).astype(np.float32),
                        np.float32([0.0, -0.0, 1.0, -1.0, 3.0e38, 1.0e-30, 1.0 + 2.0 ** -23, 255.99998, 1.00390625, 1.005859375])])
    h, m, l = split3(x)
    assert _is_bf16(h) and _is_bf16(m) and _is_bf16(l)                 # every plane is representable in bf16 as it stands
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)), x.astype(np.float64))
    # planes shrink by 2^-8 each: what makes the dropped products negligible
    assert np.all(np.abs(m) <= np.abs(h) * 2.0 ** -8) and np.all(np.abs(l) <= np.abs(h) * 2.0 ** -16)
    # below ~1e-33 the residuals become fp32 denormals and the split is no longer exact (documented in the kernel header)
    t = np.float32([1.2e-38])
    th, tm, tl = split3(t)
    assert abs(float(th[0]) + float(tm[0]) + float(tl[0]) - float(t[0])) <= 2.0 ** -8 * float(t[0])


def test_six_terms_reproduce_the_product():
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(100000) * 10.0 ** rng.integers(-6, 6, 100000)).astype(np.float32)
    b = (rng.standard_normal(100000) * 10.0 ** rng.integers(-6, 6, 100000)).astype(np.float32)
    ah, am, al = [v.astype(np.float64) for v in split3(a)]
    bh, bm, bl = [v.astype(np.float64) for v in split3(b)]
    six = ah * bl + ah * bm + ah * bh + am * bm + am * bh + al * bh
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert float(rel.max()) <= 2.0 ** -23 * 1.01, float(rel.max())
    # a 16-row dot product accumulated in fp32 (what one MFMA does) is as close to the exact sum as the plain fp32 one
    A, B = a[:96000].reshape(-1, 16), b[:96000].reshape(-1, 16)
    ex = (A.astype(np.float64) * B.astype(np.float64)).sum(1)
    plain = np.zeros(A.shape[0], np.float32)
    for k in range(16):
        plain = (plain + A[:, k] * B[:, k]).astype(np.float32)
    s6 = six[:96000].reshape(-1, 16)
    split = np.zeros(A.shape[0], np.float32)
    for k in range(16):
        split = (split + s6[:, k].astype(np.float32)).astype(np.float32)
    scale = np.abs(A.astype(np.float64) * B.astype(np.float64)).sum(1)
    assert float((np.abs(split - ex) / scale).max()) <= 2.0 * float((np.abs(plain - ex) / scale).max()) + 2.0 ** -23
