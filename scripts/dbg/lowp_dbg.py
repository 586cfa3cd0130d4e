import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
from bnn_chaos_model_amd import ops
from oracle import lowp
np.set_printoptions(linewidth=200, precision=5)
z = np.load("tests/golden/case_swagfast_v50_0_slow.npz")
x = np.load("tests/golden/inputs.npz")["x_slow"]
B = x.shape[0]
eps = np.zeros((1, B, 2, 20), np.float32)
d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
for prec, ns in (("bf16", 1), ("bf16x3", 2), ("bf16x6", 3)):
    out, pre, summ = ops.forward(d(x), d(z["w"][None]), eps=d(eps), debug=True, precision=prec)
    got = summ.cpu().numpy()[0]
    want = lowp.pooled_summary(lowp.feature_nn(x, z["w"], ns))
    print(prec, "nan count per column:", np.isnan(got).sum(0))
    print(" sys0 got ", got[0])
    print(" sys0 want", want[0].astype(np.float32))
    print(" sys5 got ", got[5])
    print(" sys5 want", want[5].astype(np.float32))
