"""CPU check of the arithmetic behind the f16 two-plane GEMMs (pointnerf_amd/csrc/f16x3.h): an fp32 number x is carried as
h = f16_rtz(x), m = f16_rne(x - h); x = h + m to 2^-21 |x| (absolute 2^-24 in the f16 subnormal range), and the three
products the kernels keep (ah*bh + ah*bm + am*bh, fp32 accumulation) reproduce the fp32 product to ~2^-20.
numpy restatement of pn_split2 (v_cvt_pkrtz_f16_f32 = round toward zero, v_cvt_pk_f16_f32 = round to nearest even)."""
import numpy as np


def f16_rtz(x):
    x = np.asarray(x, np.float32)
    with np.errstate(over="ignore"):
        r = x.astype(np.float16)
    r = np.where(np.isinf(r) & np.isfinite(x), np.copysign(np.float16(65504.0), x).astype(np.float16), r)
    too_big = np.abs(r.astype(np.float32)) > np.abs(x)
    bits = r.view(np.uint16)
    return np.where(too_big, (bits - 1).astype(np.uint16), bits).view(np.float16)


def split2(x):
    x = np.asarray(x, np.float32)
    h = f16_rtz(x)
    m = (x - h.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h, m


def test_two_planes_carry_22_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(400000) * 10.0 ** rng.uniform(-3, 4, 400000)).astype(np.float32)
    x = np.clip(x, -65000, 65000)
    h, m = split2(x)
    err = np.abs(h.astype(np.float64) + m.astype(np.float64) - x.astype(np.float64))
    assert np.all(err <= np.maximum(2.0 ** -21 * np.abs(x), 2.0 ** -25)), float((err / np.abs(x)).max())
    assert np.all(np.abs(h.astype(np.float32)) <= np.abs(x))                       # round toward zero: never overflows
    assert np.all(np.abs(m.astype(np.float32)) <= 2.0 ** -10 * np.abs(x) + 2.0 ** -24)
    # tiny values: absolute accuracy of the f16 subnormal grid
    t = (rng.standard_normal(100000) * 1e-6).astype(np.float32)
    th, tm = split2(t)
    assert float(np.abs(th.astype(np.float64) + tm.astype(np.float64) - t).max()) <= 2.0 ** -25
    # the largest finite values saturate instead of becoming inf
    hh, mm = split2(np.float32([65504.0, -65504.0, 70000.0]))
    assert np.all(np.isfinite(hh.astype(np.float32)))


def test_three_products_reproduce_a_256_term_dot_product():
    rng = np.random.default_rng(1)
    A = rng.standard_normal((2000, 256)).astype(np.float32) * rng.uniform(0.01, 4.0, (2000, 1)).astype(np.float32)
    B = (rng.standard_normal((2000, 256)) * 0.08).astype(np.float32)
    ah, am = [v.astype(np.float64) for v in split2(A)]
    bh, bm = [v.astype(np.float64) for v in split2(B)]
    three = (ah * bh + ah * bm + am * bh).sum(1)
    exact = (A.astype(np.float64) * B.astype(np.float64)).sum(1)
    scale = np.abs(A.astype(np.float64) * B.astype(np.float64)).sum(1)
    plain = np.zeros(2000, np.float32)
    for k in range(256):
        plain = (plain + A[:, k] * B[:, k]).astype(np.float32)
    e3 = float((np.abs(three - exact) / scale).max())
    ep = float((np.abs(plain - exact) / scale).max())
    print("three-product error / sum|a b| = %.2e, sequential fp32 = %.2e" % (e3, ep))
    assert e3 <= 2.0 ** -20                                   # the dropped am*bm term and the planes' own rounding
    assert e3 <= 40 * ep                                      # the same class as fp32 round-off of the plain sum


def test_single_plane_weight_gradient_error_budget():
    """k_wgrad_f16 streams dY AND X as ONE f16 plane each, rounded to nearest (dY: the high plane of pn_split2_sat; X: the copy-out's
    v_pk_add_f16 of the tile's two planes = nearest f16 of the 22-bit value):  dW = dYh^T Xh.  The rounding errors are unbiased
    (|e| <= 2^-11 of the operand, rms 2^-12 / sqrt(3) .. 2^-11 / sqrt(3)) and add up like a random walk over the rows, i.e. like ~1.6e-4 x sqrt(sum t^2) per
    element (t = the products):
      * worst case, a gradient that is itself a random walk (every element of dW the sum of zero-mean products): the error is
        ~2^-12.5 of the typical element -- measured here 7e-5 (rms) / 3.3e-4 (max) of max |dW|;
      * a gradient with structure (products with a common sign, what dW of a network at BASELINE configs[1] looks like): a few 1e-6 of
        max |dW|.
    A plane rounded toward zero (the tile's high plane alone) would bias every product by -2^-12 on average: checked below.
    For scale: the fp32 reference's own MLP gradients differ from float64 by 3.4e-5 rms of max at configs[1]
    (tests/test_gpu_bench_config.py prints ours beside it), and the parity bar of tests/test_gpu_backward.py is 5e-4 of max."""
    rng = np.random.default_rng(5)
    rows = 16384
    X = np.maximum(rng.standard_normal((rows, 256)), 0.01 * rng.standard_normal((rows, 256))).astype(np.float32)
    noise = (rng.standard_normal((rows, 256)) * 10.0 ** rng.uniform(-3, 0, (rows, 1))).astype(np.float32)
    h, m = split2(X)
    x1 = (h.astype(np.float32) + m.astype(np.float32)).astype(np.float16)            # v_pk_add_f16: the exact sum, rounded once
    direct = X.astype(np.float16)
    # (differs from the nearest f16 of x itself only where the residual plane's own rounding -- 2^-25 absolute for |x| < 0.1, where m is
    #  an f16 subnormal -- carries x across a tie: by one unit in the last place, i.e. still within 2^-11 (1 + 2^-6) |x|)
    assert float(np.mean(x1 != direct)) <= 2e-2
    big_x = np.abs(X) >= 2.0 ** -14
    assert np.all(np.abs(x1.astype(np.float64) - X)[big_x] <= (2.0 ** -11 * (1 + 2.0 ** -6)) * np.abs(X)[big_x] + 2.0 ** -25)
    for name, dY, bar_rms, bar_max in (("random-walk gradient", noise, 1.1e-4, 6e-4), ("structured gradient", noise + np.float32(0.5), 7e-6, 4e-5)):
        S = 2.0 ** np.round(np.log2(16.0 / np.abs(dY).max()))
        ref = dY.astype(np.float64).T @ X.astype(np.float64)
        dyh = (dY * np.float32(S)).astype(np.float16).astype(np.float64)
        got = (dyh.T @ x1.astype(np.float64)) / S
        mx = np.abs(ref).max()
        rms, worst = float(np.sqrt(np.mean((got - ref) ** 2)) / mx), float(np.abs(got - ref).max() / mx)
        print("single-plane dY and X, %s: rms %.2e max %.2e of max|dW|" % (name, rms, worst))
        assert rms <= bar_rms and worst <= bar_max, name
    # the truncated plane is biased, the rounded one is not
    ref = (noise + np.float32(0.5)).astype(np.float64).T @ X.astype(np.float64)
    big = np.abs(ref) >= 0.5 * np.abs(ref).max()
    rel = lambda xp: float(np.mean(((noise + np.float32(0.5)).astype(np.float64).T @ xp.astype(np.float64) - ref)[big] / ref[big]))
    assert abs(rel(x1)) <= 1e-5 and rel(h) <= -5e-5, (rel(x1), rel(h))


def test_octave_range_reduction_keeps_the_angle():
    """csrc/f16x3.h pn_pe_octaves in numpy (fp32 steps, the two fmas exact-then-rounded): the angle handed to v_sin_f32 / v_cos_f32 is
    within 4e-7 rad (two roundings of 2^-25 revolutions) of x 2^f mod 2 pi for every octave and |x| up to 3000; a single-term x / 2pi (round 2) is off by up to |x 2^f| 6e-8"""
    rng = np.random.default_rng(0)
    HI, LO = np.float32(0.15915494), np.float32(6.4206382e-9)
    for span in (0.5, 10.0, 3000.0):
        x = rng.uniform(-span, span, 200000).astype(np.float32)
        for f in range(5):
            xf = (x * np.float32(2 ** f)).astype(np.float32)                     # exact
            k = np.rint((xf * HI).astype(np.float32))
            r = (xf.astype(np.float64) * np.float64(HI) - k).astype(np.float32)     # fma: one rounding
            r = (xf.astype(np.float64) * np.float64(LO) + r.astype(np.float64)).astype(np.float32)
            assert float(np.abs(r).max()) <= 0.5 + 1e-3
            true = xf.astype(np.float64) / (2 * np.pi) - k
            err = np.abs(r.astype(np.float64) - true) * 2 * np.pi
            assert float(err.max()) <= 4e-7, (span, f, float(err.max()))
            single = np.abs((xf * HI).astype(np.float32).astype(np.float64) - xf.astype(np.float64) / (2 * np.pi)) * 2 * np.pi
            if span >= 10.0 and f == 4:
                assert float(single.max()) > 10 * float(err.max())
