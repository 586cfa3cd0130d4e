#!/usr/bin/env python3
"""BASELINE.json configs[4]: precision sweep on the 5-planet shapes (x [sims, 3 trios, 100, 41] -> rows = sims * 3; 10 chunks x
samples draws, figures/multiswag_5_planet.py:295-298).  For each arithmetic: throughput (one GPU's share) and the deviation of
(mu, std) from the fp32 HIP path on the same inputs, weights and noise.  Prints one JSON line per arithmetic.

    python scripts/lowp_sweep.py [--sims 125000] [--samples 100]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from bnn_chaos_model_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--sims", type=int, default=125_000, help="5-planet systems on this GPU (configs[4]: 1M over 8 GPUs)")
ap.add_argument("--samples", type=int, default=100)
ap.add_argument("--chunks", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda")
B = args.sims * 3
x = bench.synthetic_x(B, dev, 11)
wa, w2, pd = bench.synthetic_ensemble(30, dev)
J = args.samples * args.chunks
idx = torch.as_tensor(np.random.default_rng(0).integers(0, 30, J).astype(np.int32)).to(dev)
W = ops.swag_draw(wa, w2, pd, idx, philox_seed=7)
ref = None
for prec in ("f32", "bf16x6", "f16x3", "bf16x3", "f16", "bf16"):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ops.forward(x, W, nchunks=args.chunks, philox_seed=7, precision=prec)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    evals = B * args.samples
    rec = {"arithmetic": prec, "evals_per_s": evals / dt, "ms": dt * 1e3, "rows": B, "samples": args.samples, "chunks": args.chunks}
    if ref is None:
        ref = out.double()
    else:
        d = (out.double() - ref).abs()
        rec.update({"max_abs_dmu": d[..., 0].max().item(), "max_abs_dstd": d[..., 1].max().item(),
                    "median_abs_dmu": d[..., 0].median().item(), "p99.9_abs_dmu": torch.quantile(d[..., 0].flatten()[:16_000_000], 0.999).item(),
                    "max_rel": (d / ref.abs()).max().item()})
    print(json.dumps(rec), flush=True)
