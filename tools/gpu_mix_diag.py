"""dev: where the device's mixed tile GEMM differs from the numpy restatement (K = 272)"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mix_case
from pointnerf_amd import _lib as L
dev = torch.device("cuda:0"); lib = L.lib()
def run(x, w, K):
    dx, dw = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    img = torch.zeros(lib.pnerf_mlp_packed_bytes(), dtype=torch.uint8, device=dev)
    out = torch.full((64, 256), float("nan"), device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.check(lib.pnerf_debug_mix_gemm(P(dw), K, P(dx), P(img), P(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "mix")
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.float64), img.cpu().numpy()
for K in (272, 288):
    x, w = mix_case.build(K)
    o, img = run(x, w, K)
    ref = mix_case.restate(x, w)
    d = o - ref
    print("K", K, "max |d| / max|ref|", np.abs(d).max() / np.abs(ref).max())
    # hypotheses on the tail
    xh = mix_case.f16(x); xm = mix_case.f16(x - xh); wh = mix_case.f16(w); wm = mix_case.f16(w - wh)
    t = slice(256, K)
    terms = {"xh.wh": xh[:, t].astype(np.float64) @ wh[:, t].astype(np.float64).T, "xh.wm": xh[:, t].astype(np.float64) @ wm[:, t].astype(np.float64).T,
             "xm.wh": xm[:, t].astype(np.float64) @ wh[:, t].astype(np.float64).T}
    for k, v in terms.items():
        for sgn in (-1, 1):
            print("   ref %+d %s: max|d| %.3e" % (sgn, k, np.abs(o - (ref + sgn * v)).max() / np.abs(ref).max()))
    r, c = np.unravel_index(np.abs(d).argmax(), d.shape)
    print("   worst at row %d feature %d; per-row max %s" % (r, c, np.array2string(np.abs(d).max(1)[:8], precision=2)))
    print("   per-feature-block max", [float("%.2e" % np.abs(d[:, 32 * b:32 * b + 32]).max()) for b in range(8)])
    # x with a zero tail
    x2 = x.copy(); x2[:, 256:] = 0
    o2, _ = run(x2, w, K)
    print("   zero tail: max|d|/max|ref|", np.abs(o2 - mix_case.restate(x2, w)).max() / np.abs(ref).max())
    # the tail image against the host split
    NT = (K - 256) // 16
    u4 = np.frombuffer(img.tobytes(), dtype=np.uint16)
    base = 4 * 8 * 8 * 64 * 8      # in uint16 units: NS MB 8 64 uint4 x 8 halves
    bad = 0
    for tc in range(NT):
        for mb in range(8):
            for plane in range(2):
                for lane in range(64):
                    off = base + (((tc * 8 + mb) * 2 + plane) * 64 + lane) * 8
                    got = u4[off:off + 8].view(np.float16).astype(np.float32)
                    m = 32 * mb + (lane & 31); k0 = 256 + 16 * tc + 8 * (lane >> 5)
                    want = (wh if plane == 0 else wm)[m, k0:k0 + 8]
                    bad += int(np.abs(got - want).max() > 0)
    print("   tail image fragments that differ from the host split:", bad)

# ---- the superchunk part of the image against a numpy packing (K = 272)
def e4m3_enc(v):
    v = np.asarray(v, np.float64); out = np.zeros(v.shape, np.uint8)
    a = np.abs(v); sg = (np.signbit(v)).astype(np.uint8) << 7
    q = mix_case.q_e4m3(v); aq = np.abs(q)
    e = np.floor(np.log2(np.maximum(aq, 2.0 ** -30))); e = np.clip(e, -6, 8)
    sub = aq < 2.0 ** -6
    m = np.where(sub, np.round(aq * 512), np.round(aq / 2.0 ** e * 8) - 8).astype(np.int64)
    E = np.where(sub, 0, e + 7).astype(np.int64)
    return (sg | (E << 3).astype(np.uint8) | m.astype(np.uint8)).astype(np.uint8)
K = 272
x, w = mix_case.build(K)
o, img = run(x, w, K)
wh = mix_case.f16(w); wm32 = (w - wh).astype(np.float32)
MB = 8
u8 = np.frombuffer(img.tobytes(), dtype=np.uint8)
sc_base = (4 * 8 + 1 * 2) * MB * 64 * 16
bad_h = bad_q = bad_s = 0
first = []
for s in range(4):
    for mb in range(MB):
        for lane in range(64):
            m = 32 * mb + (lane & 31); hf = lane >> 5
            base = (((s * MB + mb) * 8) * 64 + lane) * 16
            for r in range(4):
                k0 = 64 * s + 16 * r + 8 * hf
                got = u8[base + r * 1024: base + r * 1024 + 16].view(np.float16).astype(np.float32)
                if np.abs(got - wh[m, k0:k0 + 8]).max() > 0: bad_h += 1
            scw = int(u8[sc_base + ((s * MB + mb) * 64 + lane) * 4: sc_base + ((s * MB + mb) * 64 + lane) * 4 + 4].view(np.uint32)[0])
            for j in range(2):
                cols = np.concatenate([np.arange(8 * (8 * s + 4 * j + 2 * hf + t), 8 * (8 * s + 4 * j + 2 * hf + t) + 8) for t in range(2)])
                mx = np.float32(max(np.abs(wh[m, cols]).max(), np.abs(wm32[m, cols]).max() * np.float32(2048.0)))
                fr, ex = np.frexp(mx)
                be = int((ex - 8 if fr > 0.875 else ex - 9)) if mx > 0 else 0
                sb = (scw >> (8 * j)) & 255
                if sb != be + 127:
                    bad_s += 1
                    if len(first) < 6: first.append(("scale", s, mb, lane, j, sb, be + 127, float(mx)))
                for t in range(2):
                    got = u8[base + (4 + 2 * j + t) * 1024: base + (4 + 2 * j + t) * 1024 + 16]
                    c8 = cols[8 * t: 8 * t + 8]
                    want = np.concatenate([e4m3_enc(wm32[m, c8].astype(np.float64) * 2048.0 / 2.0 ** be), e4m3_enc(wh[m, c8].astype(np.float64) / 2.0 ** be)])
                    if (got != want).any():
                        bad_q += 1
                        if len(first) < 6: first.append(("q", s, mb, lane, j, t, got.tolist(), want.tolist()))
print("image vs numpy packing: h fragments differ %d, scales differ %d, e4m3 fragments differ %d" % (bad_h, bad_s, bad_q))
for f in first: print("   ", f)
ref = mix_case.restate(x, w)
d = np.abs(o - ref)
print("K 272: per-feature max |d| in block 4:", np.array2string(d[:, 128:160].max(0), precision=1))
f = 128 + int(d[:, 128:160].max(0).argmax())
print("worst feature", f, "row errs", np.array2string(d[:8, f], precision=2))
for s in range(4):
    for lane in ((f & 31), (f & 31) + 32):
        off = sc_base + ((s * MB + 4) * 64 + lane) * 4
        print("   s %d lane %d scale bytes %s  max|wh| %.3e max|wm|*2048 %.3e" % (s, lane, u8[off:off + 4].tolist(),
              np.abs(wh[f, 64 * s + 32 * (lane >> 5) * 0: 64 * s + 64]).max(), np.abs(wm32[f, 64 * s: 64 * s + 64]).max() * 2048))
# the same weights with feature f's row copied to feature 5 (block 0): does the error follow the data?
w2 = w.copy(); w2[5] = w[f]
o2, _ = run(x, w2, K)
d2 = np.abs(o2 - mix_case.restate(x, w2))
print("row f copied to feature 5: err at 5 %.2e, at f %.2e" % (d2[:, 5].max(), d2[:, f].max()))
