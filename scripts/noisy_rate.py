"""Throughput of forward(noisy_val=True) with all noise generated in-kernel (engine A, 41 columns), for the record."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bnn_chaos_model_amd import ops
import bench
dev = torch.device("cuda")
B, J = 10000, 300
x = bench.synthetic_x(B, dev, 1)
wa, w2, pd = bench.synthetic_ensemble(30, dev)
idx = (torch.arange(J, dtype=torch.int32) % 30).to(dev)
W = ops.swag_draw(wa, w2, pd, idx, philox_seed=1)
for noisy in (False, True):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ops.forward(x, W, philox_seed=1, noisy=noisy)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"noisy={noisy}: {B * J / (t1 - t0):.4g} evals/s ({(t1 - t0) * 1e3:.1f} ms)")
