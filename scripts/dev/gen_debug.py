"""Debug aid for the generic engine: where do its numbers leave the oracle's?  (GPU box)"""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import load_golden
from bnn_chaos_model_amd import ops
from oracle import oracle as orc
from test_hip_arch import plan_and_arch, tp, dev

np.set_printoptions(linewidth=200, precision=4, suppress=True)
names = sys.argv[1:] or ["h64l16"]
# 1. the v50 network: generic vs register-resident kernels
z0 = load_golden("swag_v50_0.npz")
x = dev(load_golden("inputs.npz")["x_slow"][:16])
W = dev(z0["w_avg"][None])
plan = ops.get_plan()
a = ops.forward(x, W, philox_seed=1, plan=plan, debug=True)
b = ops.forward(x, W, philox_seed=1, plan=plan, debug=True, engine="generic")
d = (a[2] - b[2]).abs()[0].cpu().numpy()
print("v50 summary |fast - generic| max per column:", d.max(0))
print("v50 out diff", (a[0] - b[0]).abs().max().item())
for name in names:
    z = load_golden(f"case_arch_{name}.npz")
    plan, arch = plan_and_arch(ops, orc, z)
    t = tp(z, "forward_noisy0_tape")
    eps = dev(np.stack([t[0], t[1]], 1)[None])
    out, pre, summ = ops.forward(dev(z["x"]), dev(z["swagfast_w"][None]), eps=eps, plan=plan, debug=True, engine="generic")
    o, ex = orc.forward(z["x"], z["swagfast_w"], t[0], t[1], arch=arch, sched=orc.make_schedule(None, pool_parts=4), extras=True)
    ds = np.abs(summ[0].cpu().numpy() - ex["summary"])
    print(name, "summary diff max per column:", ds.max(0))
    print(name, "pre diff", np.abs(pre[0].cpu().numpy() - ex["pre_clamp"]).max(0), "out diff", np.abs(out[0].cpu().numpy() - o).max(0))
