"""dev: per-phase timeline of k_agg_forward / k_agg_backward on the bench workload.

Needs libpnerf_hip.so built with  make -C pointnerf_amd/csrc -B EXTRA_DEFS=-DPN_PHASE_TRACE  (never the shipped build):
thread 0 of every workgroup stamps s_memrealtime (100 MHz) at each phase boundary of tile iterations 20..25 together
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


FWD_GEMM = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8)]
BWD_GEMM = [(0, 1), (1, 2), (3, 4), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10)]
FWD_NAMES = {(0, 1): "G1(A) | boundary B", (1, 2): "G1(B) | E1(A)", (2, 3): "G2(A) | E1(B), copy A", (3, 4): "G2(B) | E2(A), copy B",
             (4, 5): "G3(A) | E2(B), copy A", (5, 6): "G3(B) | E3(A), copy B", (6, 7): "G4(A) | E3(B), copy A", (7, 8): "G4(B) | copy B, boundary A",
             (0, 9): "  S1: start -> slot 1 (requests)", (9, 10): "  S1: slots 1-3", (10, 11): "  S1: E4 (4-67)", (11, 12): "  S1: alpha head (68-141)", (12, 13): "  S1: K-sums + h4 copy (142-215)",
             (13, 14): "  S1: barrier + geometry (216-223)", (14, 15): "  S1: embedding PE (224-271)", (15, 16): "  S1: distance PE + pad (272-295)", (16, 17): "  S1: barrier + weights (296-299)",
             (17, 18): "  S1: slots 300-375 (index shift only)", (18, 1): "  S1: tail (375-575) + barrier"}
BWD_NAMES = {(0, 1): "G(A,4) | boundary B", (1, 2): "G(B,4) | E(A)", (2, 3): "(between steps)", (3, 4): "G(A,3) | E(B), copy A, extras A", (4, 5): "(between steps)",
             (5, 6): "G(B,3) | E(A), copy B, extras B", (6, 7): "G(A,2) | E(B), copy A", (7, 8): "G(B,2) | E(A), copy B", (8, 9): "G(A,1) | E(B), copy A",
             (9, 10): "G(B,1) | copy B, boundary A",
             (0, 11): "  S1: start -> slot 0", (11, 12): "  S1: slots 0-3", (12, 13): "  S1: slots 3-9", (13, 14): "  S1: slots 9-17", (14, 15): "  S1: slots 17-34",
             (15, 16): "  S1: slots 34-50", (16, 17): "  S1: slots 50-74", (17, 18): "  S1: slots 74-140", (18, 19): "  S1: slots 140-243",
             (19, 20): "  S1: slots 243-319", (20, 1): "  S1: slots 319-511 + barrier"}

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
print(json.dumps({"forward": analyse(fwd, FWD_NAMES, FWD_GEMM, 8), "backward": analyse(bwd, BWD_NAMES, BWD_GEMM, 10)}, indent=1))
