"""Generate tests/golden/trajectory_envelope.npz: the yardstick of tests/test_gpu_zz_convergence.py::test_200_steps_inside_the_oracle_ensemble.

    python tests/golden/make_trajectory_envelope.py [--runs 16] [--steps 200]

The CPU oracle (oracle/pyref.py + torch.optim.Adam: the reference's loop body, models/mvs_points_volumetric_model.py:98-118) optimises the
`small_k8` case for 200 steps once unperturbed and `runs` times with every MLP weight moved by a relative +-2^-23 (half an fp32 ulp, a
different sign mask per run).  Adam normalises every gradient element to ~lr, so a last-bit difference grows step by step and the runs
decorrelate somewhere after step ~100; WHEN is chance (a pre-activation crossing a LeakyReLU kink), so the yardstick is the ENSEMBLE: per step
the largest relative distance any perturbed run has from the unperturbed one.  Stored: the unperturbed losses, every perturbed run's losses and
the per-step maximum.  Inputs are regenerated from seeds (tests/cases.py), nothing else is stored.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tests"))

from cases import build_case                        # noqa: E402
from test_gpu_train_steps import oracle_steps       # noqa: E402


def perturbed(mlp, seed):
    """every MLP weight x (1 +- 2^-23), the sign mask drawn from `seed`"""
    g = torch.Generator().manual_seed(seed)
    return {k: v * (1.0 + (torch.randint(0, 2, v.shape, generator=g).float() * 2 - 1) * 2.0 ** -23) for k, v in mlp.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=16)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", default=os.path.join(HERE, "trajectory_envelope.npz"))
    a = ap.parse_args()
    opt, xyz, attrs, inp, mlp = build_case("small_k8")
    t0 = time.time()
    ref = np.array(oracle_steps(opt, xyz, attrs, inp, mlp, a.steps)[0], dtype=np.float64)
    print("unperturbed: %.0f s, loss %.6f -> %.6f" % (time.time() - t0, ref[0], ref[-1]), flush=True)
    runs = []
    for r in range(a.runs):
        t0 = time.time()
        runs.append(np.array(oracle_steps(opt, xyz, attrs, inp, perturbed(mlp, 1000 + r), a.steps)[0], dtype=np.float64))
        d = np.abs(runs[-1] - ref) / np.maximum(np.abs(ref), 1e-6)
        print("run %2d: %.0f s, relative distance at steps 50/100/150/200: %.1e %.1e %.1e %.1e" % (r, time.time() - t0, d[49], d[99], d[min(149, a.steps - 1)], d[-1]), flush=True)
        runs_a = np.stack(runs)
        env = (np.abs(runs_a - ref) / np.maximum(np.abs(ref), 1e-6)).max(0)
        np.savez_compressed(a.out, reference=ref, perturbed=runs_a, envelope=env, seeds=np.arange(1000, 1000 + len(runs)))


if __name__ == "__main__":
    main()
