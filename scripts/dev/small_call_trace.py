"""The kernels of ONE small forward_swag_fast call on the GPU timeline (run under `rocprofv3 --kernel-trace`): the scripts' shapes --
15 rows (figures/multiswag_5_planet.py:295-298 chunks) and 3 000 rows (figures/main_figures.py:154-156) -- each call followed by a
synchronisation, as the scripts' `.cpu()` does, so that the trace shows one call's chain with nothing queued behind it.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_trace_small -- python scripts/dev/small_call_trace.py"""
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
for B in (15, 3000):
    x = bench.synthetic_x(B, torch.device("cuda"), 1)
    z1, z2, eps = torch.randn(1, 7583, device="cuda"), torch.randn(1, 30, device="cuda"), torch.randn(1, B, 2, 20, device="cuda")
    for _ in range(30):
        ops.multiswag(x, wa, w2, pd, idx, z1, z2, eps)
        torch.cuda.synchronize()
print("done")
