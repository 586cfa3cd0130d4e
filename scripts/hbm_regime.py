#!/usr/bin/env python3
"""Where does the path meet the HBM roof?  With many draws per system x is re-read on-die (L2) and every arithmetic is bound by its
instruction issue.  With ONE draw per system each row of x crosses HBM exactly once per evaluation: this
script times 1M systems (x = 16.4 GB) under J = 1, 2, 4, 8 draws in fp32 and in the opt-in bf16 form and prints the algorithmic
HBM rate (16 408 B per eval) against the 8 TB/s peak.    python scripts/hbm_regime.py
"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from bnn_chaos_model_amd import ops

dev = torch.device("cuda")
B = 1_000_000
x = bench.synthetic_x(B, dev, 5)
wa, w2, pd = bench.synthetic_ensemble(30, dev)
for prec in ("f32", "bf16"):
    for J in (1, 2, 4, 8):
        idx = (torch.arange(J, dtype=torch.int32) % 30).to(dev)
        W = ops.swag_draw(wa, w2, pd, idx, philox_seed=1)
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = ops.forward(x, W, philox_seed=1, precision=prec)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        evals = B * J
        print(json.dumps({"arithmetic": prec, "draws_per_system": J, "ms": best * 1e3, "evals_per_s": evals / best,
                          "hbm_algorithmic_GBs": evals * 16408 / best / 1e9, "frac_of_8TBs": evals * 16408 / best / 8e12,
                          "x_GBs_if_read_once": B * 16400 / best / 1e9}), flush=True)
