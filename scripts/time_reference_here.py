#!/usr/bin/env python3
"""Times the UNMODIFIED reference (imported through tests/golden/make_golden.py's stub) in the build container, on the shapes
bench.py's CPU baselines use: configs[0] (1 000 systems, one draw) and the per-call cost of sample_weights.  Runs only where
/root/reference exists; the numbers are quoted in DESIGN.md section 5 next to the GPU box's cpu_baseline figures.

    python scripts/time_reference_here.py [--systems 1000] [--calls 5]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--systems", type=int, default=1000)
ap.add_argument("--calls", type=int, default=5)
a = ap.parse_args()

ref = make_golden.import_reference()
model = ref.load_swag(make_golden.pretrained(0)).cpu().eval()
g = torch.Generator().manual_seed(123)
B = a.systems
x = torch.randn(B, 1, 41, generator=g) + 0.1 * torch.randn(B, 100, 41, generator=g)
x[:, :, 0] = torch.linspace(-1.71, 1.74, 100)[None]
threads = torch.get_num_threads()

with torch.no_grad():
    model.forward_swag_fast(x[:8], scale=0.5)  # warm-up
    t_draw, t_fwd, t_call = [], [], []
    for _ in range(a.calls):
        t0 = time.perf_counter(); model.sample_weights(scale=0.5); t1 = time.perf_counter()
        model(x, noisy_val=False); t2 = time.perf_counter()
        model.forward_swag_fast(x, scale=0.5); t3 = time.perf_counter()
        t_draw.append(t1 - t0); t_fwd.append(t2 - t1); t_call.append(t3 - t2)
print(f"reference on {threads} torch threads, {B} systems x 1 draw (median of {a.calls}):")
print(f"  sample_weights          {np.median(t_draw) * 1e3:9.1f} ms per draw")
print(f"  forward (loaded weights){np.median(t_fwd) * 1e3:9.1f} ms  = {B / np.median(t_fwd):10.0f} evals/s")
print(f"  forward_swag_fast       {np.median(t_call) * 1e3:9.1f} ms  = {B / np.median(t_call):10.0f} evals/s (one draw + one forward)")
