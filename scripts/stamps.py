"""Reads the per-phase s_memtime sums of a BNN_STAMPS build of the 4x4x1 kernel (diagnostic only)."""
import os, sys
os.environ["BNN_CHAOS_SO"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bnn_chaos_model_amd", "csrc", "libbnn_stamps.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from bnn_chaos_model_amd import ops
import bench
B, J = 10000, 600
dev = torch.device("cuda")
x = bench.synthetic_x(B, dev, 1)
wa, w2, pd = bench.synthetic_ensemble(30, dev)
idx = (torch.arange(J, dtype=torch.int32) % 30).to(dev)
W = ops.swag_draw(wa, w2, pd, idx, philox_seed=1)
for rep in range(2):
    out, pre, summ = ops.forward(x, W, philox_seed=1, debug=True)
torch.cuda.synchronize()
spc = 512
nblk = J * ((B + spc - 1) // spc)
st = pre.view(torch.int64).flatten()[: nblk * 12].view(nblk, 12).cpu().numpy().astype(np.float64)
names = ["prologue", "batch setup", "layer1", "relu1+loads", "layer2", "relu2", "layer3", "pool", "merge+finish", "regress_nn+store", "", ""]
tot = st.sum(1)
full = st[tot > np.percentile(tot, 50)]  # blocks with a full 512 systems
m = full.mean(0)
print("cycles per workgroup (wave 0), mean over full blocks: total %.0f" % m.sum())
tiles = 25 * 8  # 512 systems / 4 waves / 16 systems per batch = 8 batches of 25 tiles
for n, v in zip(names, m):
    if n:
        print(f"  {n:18s} {v:12.0f}  {100 * v / m.sum():5.1f} %   per tile {v / tiles:8.1f}")
