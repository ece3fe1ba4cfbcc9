"""dev: per-phase timeline of k_agg_forward / k_agg_backward on the bench workload.

Needs libpnerf_hip.so built with  make -C pointnerf_amd/csrc -B EXTRA_DEFS=-DPN_PHASE_TRACE  (never the shipped build):
thread 0 of every workgroup stamps s_memrealtime (100 MHz) at each phase boundary of tile iterations 20..25 (one 64-row tile per iteration, two workgroups per CU) together
with HW_ID / XCC_ID, so that the two workgroups sharing a CU can be paired and their GEMM phases overlaid.
Prints a JSON summary; raw stamps go to gpurun_out/phase_trace.npz."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pointnerf_amd import config, _lib, dist as pdist

WGS, ITERS, SLOTS = 512, 6, 24
dev = torch.device("cuda:0")
opt = config.bench_lego_opt(is_train=1)
model = bench.build_model(opt, 2_000_000, dev)
agg, npnt = model.aggregator, model.neural_points
params = [p for p in agg.parameters()] + [npnt.points_embeding, npnt.points_conf, npnt.points_dir, npnt.points_color]
for i in range(3):
    for p in params:
        p.grad = None
    inp = bench.step_inputs(i, 0, 1, 65536, dev)
    out = model(**inp)
    loss = pdist.hot_path_loss(opt, out, inp["gt_image"])
    loss.backward()
torch.cuda.synchronize()
lib = _lib.lib()


def read(name):
    buf = np.zeros(WGS * ITERS * SLOTS, dtype=np.uint64)
    fn = getattr(lib, name)
    fn.restype = ctypes.c_int
    rc = fn(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
    assert rc == 0, rc
    return buf.reshape(WGS, ITERS, SLOTS)


FWD_GEMM = [(2, 3), (5, 6), (8, 9), (11, 12)]
BWD_GEMM = [(4, 5), (7, 8), (10, 11), (13, 14)]
FWD_NAMES = {(0, 1): "build X0 + weights", (1, 2): "acc = bias", (2, 3): "GEMM1 + copy-out X0 cols 224..287 + barrier", (3, 4): "E1 + barrier", (4, 5): "acc = bias",
             (5, 6): "GEMM2 + copy-out h1 + barrier", (6, 7): "E2 + extras + barrier", (7, 8): "acc = bias", (8, 9): "GEMM3 + copy-out h2x + barrier", (9, 10): "E3 + barrier",
             (10, 11): "acc = bias", (11, 12): "GEMM4 + copy-out h3", (12, 13): "next gather issue + barrier", (13, 14): "E4 + barrier",
             (14, 15): "tail in one pass: alpha + h4 copy + K-sums + sigma", (15, 16): "-"}
BWD_NAMES = {(0, 1): "load h4 / meta / d f + barrier", (1, 2): "front in one pass: alpha backward + dY4", (2, 3): "barrier", (3, 4): "acc zero", (4, 5): "GEMM4 + copy-out dY4 + barrier",
             (5, 6): "E(m3) + barrier", (6, 7): "acc zero", (7, 8): "GEMM3 + extras + copy-out dY3 + barrier", (8, 9): "E(m2) + extras finish + barrier", (9, 10): "acc zero",
             (10, 11): "GEMM2 + copy-out dY2 + barrier", (11, 12): "E(m1) + barrier", (12, 13): "acc zero", (13, 14): "GEMM1 + copy-out dY1 + barrier", (14, 15): "dX0 -> LDS + barrier",
             (15, 16): "embedding gradient"}

def analyse(tr, names, gemm, last):
    t = tr[:, :, :max(last, max(max(k) for k in names)) + 1].astype(np.int64)
    ok = (t[:, :, 0] > 0).all(axis=1) & (t[:, :, last] > 0).all(axis=1)
    t = t[ok]
    hw = tr[ok, 0, SLOTS - 1]
    res = {"workgroups_traced": int(ok.sum())}
    dur = {}
    for (a, b), n in names.items():
        d = (t[:, :, b] - t[:, :, a]).astype(np.float64) * 0.01      # us
        dur["%02d-%02d %s" % (a, b, n)] = [round(float(d.mean()), 2), round(float(np.percentile(d, 90)), 2)]
    res["phase_us_mean_p90"] = dur
    it = (t[:, 1:, 0] - t[:, :-1, 0]).astype(np.float64) * 0.01
    res["tile_iteration_us_mean"] = round(float(it.mean()), 2)       # backward: one iteration = a PAIR of tiles
    res["gemm_us_per_tile"] = round(float(sum((t[:, :, b] - t[:, :, a]).mean() for a, b in gemm) * 0.01), 2)
    # pair workgroups by CU: xcc id (high word) + HW_ID bits 8..15 (cu, sh, se)
    key = ((hw >> np.uint64(32)) & np.uint64(0xF)) * np.uint64(65536) + (hw & np.uint64(0xFF00))
    both = one = none = 0.0
    pairs = 0
    for k in np.unique(key):
        idx = np.nonzero(key == k)[0]
        if len(idx) != 2:
            continue
        a, b = t[idx[0]], t[idx[1]]
        lo, hi = max(a[0, 0], b[0, 0]), min(a[-1, last], b[-1, last])
        if hi <= lo:
            continue
        ev = []
        for w in (a, b):
            for i in range(w.shape[0]):
                for (s, e) in gemm:
                    x0, x1 = max(lo, min(hi, w[i, s])), max(lo, min(hi, w[i, e]))
                    if x1 > x0:
                        ev.append((x0, 1)); ev.append((x1, -1))
        ev.sort()
        cur, prev = 0, lo
        acc = [0, 0, 0]
        for x, dlt in ev:
            acc[cur] += x - prev
            prev, cur = x, cur + dlt
        acc[cur] += hi - prev
        tot = float(hi - lo)
        none += acc[0] / tot; one += acc[1] / tot; both += acc[2] / tot
        pairs += 1
    if pairs:
        res["cu_pairs"] = pairs
        res["time_frac_in_gemm_phase"] = {"neither_wg": round(none / pairs, 3), "one_wg": round(one / pairs, 3), "both_wgs": round(both / pairs, 3)}
    return res


fwd, bwd = read("pnerf_debug_trace_fwd"), read("pnerf_debug_trace_bwd")
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/phase_trace.npz", fwd=fwd, bwd=bwd)
print(json.dumps({"forward": analyse(fwd, FWD_NAMES, FWD_GEMM, 16), "backward": analyse(bwd, BWD_NAMES, BWD_GEMM, 16)}, indent=1))
