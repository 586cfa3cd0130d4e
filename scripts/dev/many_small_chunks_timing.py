"""The 5-planet loop as ONE call (figures/multiswag_5_planet.py:295-298 through sample_full_swag_many): 150 rows in 10 chunks of 15 under
`samples` x 10 draws, in-kernel Philox -- the library's default launch form against the plain form (systems_per_block=64)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bnn_chaos_model_amd import ops  # noqa: E402
import bench  # noqa: E402

z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "ensemble_v50.npz"))
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
wa, w2, pd = dev(z["w_avg"]), dev(z["w2_avg"]), dev(z["pre_D"])


def per_call(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B, nch, samples in ((150, 10, 100), (150, 10, 1000), (15000, 1000, 10), (48, 3, 1000)):
    x = bench.synthetic_x(B, torch.device("cuda"), 1)
    J = samples * nch
    idx = torch.as_tensor(np.random.default_rng(0).integers(0, 30, J).astype(np.int32)).cuda()
    out = torch.empty((samples, B, 2), device="cuda")
    row = {"rows": B, "chunks": nch, "samples": samples, "evals": B * samples}
    for single in (True, False):
        a = ops.multiswag(x, wa, w2, pd, idx, nchunks=nch, philox_seed=3, single_launch=single)
        b = ops.multiswag(x, wa, w2, pd, idx, nchunks=nch, philox_seed=3, single_launch=single, systems_per_block=64)
        assert torch.equal(a, b)
        tag = "fused" if single else "workspace"
        row[f"default_{tag}_us"] = per_call(lambda: ops.multiswag(x, wa, w2, pd, idx, nchunks=nch, philox_seed=3, single_launch=single, out=out, assume_finite=True))
        row[f"plain_{tag}_us"] = per_call(lambda: ops.multiswag(x, wa, w2, pd, idx, nchunks=nch, philox_seed=3, single_launch=single, out=out, assume_finite=True, systems_per_block=64))
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in row.items()}, flush=True)
