#!/usr/bin/env python3
"""Largest relative deviation of the HIP path from the reference's captured outputs over the forward fixtures (bar: 1e-5)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, tape  # noqa: E402
from bnn_chaos_model_amd import ops  # noqa: E402

dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
inputs = load_golden("inputs.npz")
worst = 0.0
for si in (0, 12):
    for xname in ("slow", "iid", "const4"):
        z = load_golden(f"case_swagfast_v50_{si}_{xname}.npz")
        tp = tape(z)
        eps = np.stack([tp[2][1], tp[3][1]], axis=1)[None]
        out = ops.forward(dev(inputs[f"x_{xname}"]), dev(z["w"][None]), eps=dev(eps))[0].cpu().numpy()
        rel = np.abs(out - z["out"]) / np.abs(z["out"])
        worst = max(worst, rel.max())
        print(f"swagfast v50_{si} {xname:7s} max rel {rel.max():.2e}")
print(f"worst {worst:.2e} (bar 1e-5)")
