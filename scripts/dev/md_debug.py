import sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import load_golden
from bnn_chaos_model_amd import ops
z = load_golden("swag_v50_0.npz")
wa, w2, pd = (torch.tensor(z[k][None]).cuda() for k in ("w_avg", "w2_avg", "pre_D"))
X = torch.tensor(np.tile(load_golden("inputs.npz")["x_slow"], (3, 1, 1))[:77]).cuda()
idx = torch.zeros(40, dtype=torch.int32)
full = ops.multiswag(X, wa, w2, pd, idx, nchunks=10, philox_seed=1)
for sl in (True, False):
    for lo, hi in ((0, 26), (26, 52), (52, 77)):
        o = torch.full((4, hi - lo, 2), -1.0, device="cuda")
        ops.multiswag(X[lo:hi].contiguous(), wa, w2, pd, idx, nchunks=10, philox_seed=1, system_id0=lo, chunk_B=77, chunk_off=lo, out=o, single_launch=sl)
        bad = (o != full[:, lo:hi]).any(2)
        print(sl, lo, hi, "mismatch rows per sample:", [np.nonzero(b.cpu().numpy())[0].tolist() for b in bad], "unwritten:", int((o == -1).sum()))
