"""One forward of the 5-planet share in one arithmetic (f32 | bf16 | bf16x3 | bf16x6), for profiling runs (scripts/lowp_pmc.sh).
   usage: python scripts/lowp_one.py [precision] [nchunks]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from bnn_chaos_model_amd import ops
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
B = 375000
x = bench.synthetic_x(B, dev, 11)
wa, w2, pd = bench.synthetic_ensemble(30, dev)
J = 100 * nch
idx = torch.as_tensor(np.random.default_rng(0).integers(0, 30, J).astype(np.int32)).to(dev)
W = ops.swag_draw(wa, w2, pd, idx, philox_seed=7)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ops.forward(x, W, nchunks=nch, philox_seed=7, precision=prec)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(prec, nch, "%.4g evals/s" % (B * 100 / dt), "%.2f ms" % (dt * 1e3))
