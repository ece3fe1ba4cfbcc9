"""Does optimisation with the shipped weight-gradient arithmetic (ONE f16 plane per operand, 11-bit significands, fp32 accumulation:
DESIGN 4.1) behave like optimisation with fp32-class weight gradients?  The reference's loop runs 200 000 steps
(run/train_ft.py:829-937 around models/mvs_points_volumetric_model.py:98-118); round 3's longest check was 3 steps.

* 200 steps of the oracle's small case against the fp32 CPU oracle with torch.optim.Adam: deterministic to 1e-4 for the first 75 steps, inside the
  envelope of an ensemble of perturbed oracle runs afterwards (tests/golden/trajectory_envelope.npz).
* 2 000 steps of a teacher / student problem (tests/convergence_case.py) with one plane and with two planes per operand
  (ops.set_wgrad_planes), identical batches: final loss and held-out PSNR of the two arithmetics must agree within the spread that
  repeated runs of ONE arithmetic show (the backward's atomics make no two runs bit-identical)."""
import json
import os

import numpy as np
import pytest
import torch

import convergence_case as C
from pointnerf_amd import ops
from cases import build_case
from test_gpu_train_steps import device_steps, oracle_steps

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory_envelope.npz")
EARLY, LOOKAHEAD = 100, 25          # (the fixture's envelope stays below 1e-4 / 3 up to step 107)
LATE_CAP = 0.15                     # no assertion ever tolerates more than this relative loss difference (3 x the final envelope is 0.144)


def envelope_bar(env):
    """the bar of step t: 3 x the largest distance ANY member of the perturbed-oracle ensemble has shown up to step t + LOOKAHEAD, at least 1e-4.
    (The look-ahead is what keeps the bar independent of WHEN a run decorrelates: a run that leaves the common trajectory 25 steps before the
    earliest of the 16 ensemble members is still inside.)"""
    run_max = np.maximum.accumulate(env)
    ahead = np.concatenate([run_max[LOOKAHEAD:], np.full(LOOKAHEAD, run_max[-1])])
    return np.minimum(np.maximum(3.0 * ahead, 1e-4), LATE_CAP)


def test_200_steps_inside_the_oracle_ensemble():
    """200 optimisation steps of the small case on the device against the fp32 CPU oracle (torch.optim.Adam around oracle/pyref.py: the reference's loop body,
    models/mvs_points_volumetric_model.py:98-118).  Two fp32 evaluations of this optimisation do not stay together for 200 steps: Adam normalises every gradient
    element to ~lr, so a last-bit difference grows step by step, and the step at which a run leaves the common trajectory is chance (a pre-activation crossing
    a LeakyReLU kink).  No assertion here depends on when that happens:

    (i)  steps 1 .. 100 are deterministic to rounding: the device loss is within 1e-4 relative of the oracle's (measured <= 3e-6 on every box seen: 30 x
         margin), against the oracle run live on this box AND against the committed trajectory (which pins the fixture to this box's oracle);
    (ii) over all 200 steps the device stays inside 3 x the ENVELOPE of an ensemble of 16 perturbed oracle runs (every MLP weight x (1 +- 2^-23), a
         different sign mask per run; tests/golden/make_trajectory_envelope.py), with a 25-step look-ahead -- for the shipped one-plane weight gradients and
         for the two-plane (fp32-class) ones alike."""
    fx = np.load(GOLDEN)
    ref, env = fx["reference"], fx["envelope"]
    assert ref.shape == (200,) and fx["perturbed"].shape[0] >= 8
    case = build_case("small_k8")
    live, _, _ = oracle_steps(*case, EARLY)
    ours, _, _ = device_steps(*case, 200)
    old = ops.set_wgrad_planes(2)
    try:
        ours2, _, _ = device_steps(*case, 200)
    finally:
        ops.set_wgrad_planes(old)
    rel = lambda a, b: np.abs(np.asarray(a, dtype=np.float64) - b[: len(a)]) / np.maximum(np.abs(b[: len(a)]), 1e-6)
    bar = envelope_bar(env)
    d_live, d_dev, d_dev2 = rel(live, ref), rel(ours, ref), rel(ours2, ref)
    for t in (1, 10, 25, 50, 75, 100, 125, 150, 175, 200):
        print("step %3d  loss device %.6f (two-plane %.6f) oracle %.6f   relative to the oracle: device %.1e, two-plane device %.1e, ensemble envelope %.1e, bar %.1e"
              % (t, ours[t - 1], ours2[t - 1], ref[t - 1], d_dev[t - 1], d_dev2[t - 1], env[t - 1], bar[t - 1]))
    assert ref[-1] < 0.5 * ref[0], "the case must actually optimise"
    assert d_live.max() <= 1e-4, "this box's oracle differs from the committed trajectory within %d steps: %g" % (EARLY, d_live.max())
    for name, d, tr in (("one plane", d_dev, ours), ("two planes", d_dev2, ours2)):
        assert d[:EARLY].max() <= 1e-4, (name, float(d[:EARLY].max()))
        assert rel(tr[:EARLY], np.asarray(live, dtype=np.float64)).max() <= 1e-4, name
        assert (d <= bar).all(), (name, int(np.argmax(d / bar)), float((d / bar).max()))


STEPS = 2000


def test_convergence_one_plane_vs_two_planes():
    """6 + 6 runs of 2 000 steps (profiles/r04_convergence_ab.json: 12 + 12).  Every run ends at a slightly different point (the backward's atomics order the point-gradient sums
    differently from launch to launch, and the optimisation amplifies that), so the two arithmetics are compared as two samples: the
    difference of their means against the standard error of that difference (Welch), for the held-out PSNR and for the loss over ALL
    training rays evaluated after the last step (not the noisy mini-batch losses)."""
    sc = C.scene()
    runs = {1: [C.run(DEV, STEPS, 1, sc=sc) for _ in range(6)], 2: [C.run(DEV, STEPS, 2, sc=sc) for _ in range(6)]}
    stat = lambda key, planes: np.array([r[key] for r in runs[planes]])
    out = {"steps": STEPS, "psnr_heldout_before": runs[1][0]["psnr_heldout_before"], "train_mse_before": runs[1][0]["train_mse_before"]}
    for key in ("train_mse", "psnr_heldout", "psnr_train"):
        a, b = stat(key, 1), stat(key, 2)
        se = float(np.sqrt(a.var(ddof=1) / a.size + b.var(ddof=1) / b.size))
        out[key] = dict(one_plane=a.tolist(), two_planes=b.tolist(), mean_one_plane=float(a.mean()), mean_two_planes=float(b.mean()),
                        difference_of_means=float(abs(a.mean() - b.mean())), standard_error_of_the_difference=se,
                        run_to_run_std=[float(a.std(ddof=1)), float(b.std(ddof=1))])
        print(key, out[key])
    out["loss_curves"] = {"one_plane": runs[1][0]["loss_curve"], "two_planes": runs[2][0]["loss_curve"]}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/convergence_ab.json", "w") as fh:
        json.dump(out, fh)
    # the problem is a real one: every run gains > 8 dB on views it never trained on (measured over 24 + 12 runs: mean + 14 dB, run-to-run standard
    # deviation 1.0 .. 2.0 dB, worst + 11.0 -- a + 10 dB bar would fail one suite run in ten by chance alone)
    assert min(stat("psnr_heldout", 1).min(), stat("psnr_heldout", 2).min()) > out["psnr_heldout_before"] + 8.0
    # the two arithmetics are one population: |difference of means| <= 4 standard errors (floors: 0.15 dB, 5 % of the loss).  With 6 + 6 runs
    # Welch's statistic has ~10 degrees of freedom: 3 standard errors is exceeded by chance in 1.3 % of the comparisons, i.e. by one of the three
    # metrics in ~3 % of the suite runs (the round-end run stops at the first failure); 4 in 0.25 %.  The 12 + 12 runs of
    # profiles/r04_convergence_ab.json sit at 0.35 (held-out PSNR), 1.8 (training PSNR) and 1.9 (training MSE, in favour of one plane).
    for key, floor in (("psnr_heldout", 0.15), ("psnr_train", 0.15), ("train_mse", 0.05 * out["train_mse"]["mean_two_planes"])):
        o = out[key]
        assert o["difference_of_means"] <= max(4.0 * o["standard_error_of_the_difference"], floor), (key, o)
