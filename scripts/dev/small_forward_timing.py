"""Forward kernel alone (assume_finite: no scan / fix-up launches), 15 systems, series lengths 8 .. 100: time per launch from a graph of 20
launches replayed 20 times -- the slope over the number of 4-tile rounds is a round's cost, the intercept the launch + prologue + tail."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bnn_chaos_model_amd import ops  # noqa: E402
import bench  # noqa: E402

z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "swag_v50_0.npz"))
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
wa, w2, pd = dev(z["w_avg"][None]), dev(z["w2_avg"][None]), dev(z["pre_D"][None])
idx = torch.zeros(1, dtype=torch.int32, device="cuda")
W = ops.swag_draw(wa, w2, pd, idx, philox_seed=1)


def per_launch(fn, n=20, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


for B in (15, 3000):
    for T in (8, 24, 40, 56, 72, 100):
        x = bench.synthetic_x(B, torch.device("cuda"), 1)[:, :T].contiguous()
        out = torch.empty((1, B, 2), device="cuda")
        row = {"B": B, "T": T, "rounds": (T // 4 + 3) // 4}
        row["draw_kernel_us"] = per_launch(lambda: ops.swag_draw(wa, w2, pd, idx, philox_seed=1))
        zz1, zz2 = torch.randn(1, 7583, device="cuda"), torch.randn(1, 30, device="cuda")
        row["draw_kernel_explicit_z_us"] = per_launch(lambda: ops.swag_draw(wa, w2, pd, idx, zz1, zz2))
        row["workspace_small_us"] = per_launch(lambda: ops.multiswag(x, wa, w2, pd, idx, zz1, zz2, philox_seed=3, assume_finite=True, single_launch=False, out=out))
        row["fused_small_explicit_z_us"] = per_launch(lambda: ops.multiswag(x, wa, w2, pd, idx, zz1, zz2, philox_seed=3, assume_finite=True, single_launch=True, out=out))
        row["forward_small_us"] = per_launch(lambda: ops.forward(x, W, philox_seed=3, assume_finite=True))
        row["forward_plain_us"] = per_launch(lambda: ops.forward(x, W, philox_seed=3, assume_finite=True, systems_per_block=64))
        row["fused_small_us"] = per_launch(lambda: ops.multiswag(x, wa, w2, pd, idx, philox_seed=3, assume_finite=True, single_launch=True, out=out))
        row["fused_plain_us"] = per_launch(lambda: ops.multiswag(x, wa, w2, pd, idx, philox_seed=3, assume_finite=True, single_launch=True, out=out, systems_per_block=64))
        print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in row.items()}, flush=True)
