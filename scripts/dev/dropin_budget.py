"""Per-call host budget of the UNCHANGED scripts' route: FeatureRegressor.sample_full_swag called per chunk per sample
(figures/multiswag_5_planet.py:295-298: 15-row chunks; figures/main_figures.py:154-156: 3 000-row batches, 2 000 calls each).

Prints, for the model on the GPU and in host memory and for both shapes: the whole call with the scripts' `.detach().cpu()` behind it, the
same without the copy-back (enqueue cost only), and stand-alone timings of the pieces the call is made of -- so that the table says where
a call's time goes without a profiler's own overhead in it.  JSON lines go to gpurun_out/r06_dropin_budget.jsonl.

    python scripts/dev/dropin_budget.py [tag]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from bnn_chaos_model_amd import checkpoint, ops  # noqa: E402
from bnn_chaos_model_amd.regression import FeatureRegressor  # noqa: E402
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""
gold = os.path.join(ROOT, "tests", "golden")
d = tempfile.mkdtemp()
for i in (0, 12):
    z = np.load(os.path.join(gold, f"swag_v50_{i}.npz"))
    checkpoint.write_swag_file(os.path.join(d, f"m_v50_{i:02d}_output.pkl"), json.loads(str(z["hparams_json"])), json.loads(str(z["swa_params_json"])),
                               torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"]))


def timeit(fn, n=400, sync=True):
    """microseconds per call over n back-to-back calls (with sync=True the GPU's work is inside the window: GPU-bound if it is the slower side)"""
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def host_only(fn, burst=24, reps=15):
    """HOST cost per call: bursts of `burst` calls issued into an idle queue (nothing waits on the GPU inside the window), median over reps"""
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(burst):
            fn()
        ts.append((time.perf_counter() - t0) / burst * 1e6)
        torch.cuda.synchronize()
    return sorted(ts)[len(ts) // 2]


rows = []
for cuda in (True, False):
    model = FeatureRegressor(cuda=cuda, filebase=os.path.join(d, "*v50*output.pkl"), sort=True)
    for shape, B in (("5-planet chunk", 15), ("main_figures batch", 3000)):
        X = bench.synthetic_x(B, torch.device("cuda"), 1)
        if not cuda:
            X = X.cpu()
        r = {"tag": tag, "model_on_gpu": cuda, "shape": shape, "rows": B}
        r["call_plus_cpu_copy_us"] = timeit(lambda: model.sample_full_swag(X).detach().cpu(), sync=False)
        r["call_back_to_back_us"] = timeit(lambda: model.sample_full_swag(X))
        r["call_host_only_us"] = host_only(lambda: model.sample_full_swag(X))
        m = model.swag_ensemble[0]
        if cuda:
            m.cuda()
        # the pieces, stand-alone (same objects the call uses)
        r["randint_us"] = timeit(lambda: np.random.randint(0, 2), sync=False)
        r["state_gpu_us"] = timeit(lambda: m._state_gpu(), sync=False)
        r["x_to_gpu_us"] = timeit(lambda: X.detach().to("cuda", torch.float32).contiguous())
        dev_in = X.device
        r["noise_draws_us"] = timeit(lambda: (torch.randn((1, 7583), device=m._device), torch.randn((30, 1), device=m._device),
                                              torch.randn(B, 20, device=dev_in), torch.randn(B, 20, device=dev_in)))
        wa, w2, pd = m._state_gpu()
        xg = X.detach().to("cuda", torch.float32).contiguous()
        idx = torch.zeros(1, dtype=torch.int32, device="cuda")
        z1, z2, eps = torch.randn(1, 7583, device="cuda"), torch.randn(1, 30, device="cuda"), torch.randn(1, B, 2, 20, device="cuda")
        mask, lowest, net = m._op_args()
        r["custom_op_host_us"] = host_only(lambda: torch.ops.bnn_chaos.multiswag(xg, wa, w2, pd, idx, z1, z2, eps, 1, 0.5, 0, 0, 0, mask, lowest, net, False))
        r["ops_multiswag_host_us"] = host_only(lambda: ops.multiswag(xg, wa, w2, pd, idx, z1, z2, eps, plan=m._plan()))
        r["ops_multiswag_assume_finite_host_us"] = host_only(lambda: ops.multiswag(xg, wa, w2, pd, idx, z1, z2, eps, plan=m._plan(), assume_finite=True))
        r["noise_draws_host_us"] = host_only(lambda: (torch.randn((1, 7583), device=m._device), torch.randn((30, 1), device=m._device),
                                                      torch.randn(B, 20, device=dev_in), torch.randn(B, 20, device=dev_in)))
        r["forward_swag_fast_host_us"] = host_only(lambda: m.forward_swag_fast(X, scale=0.5))
        out = ops.multiswag(xg, wa, w2, pd, idx, z1, z2, eps, plan=m._plan())
        r["result_to_cpu_us"] = timeit(lambda: out[0].detach().cpu(), sync=False)
        # GPU time of one call's kernels, from events around 50 back-to-back eager calls
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ops.multiswag(xg, wa, w2, pd, idx, z1, z2, eps, plan=m._plan())
        g.replay()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(50):
            g.replay()
        ev1.record()
        torch.cuda.synchronize()
        r["gpu_chain_us_graph_replay"] = ev0.elapsed_time(ev1) / 50 * 1e3
        rows.append(r)
        print(json.dumps(r), flush=True)
        if cuda:
            m.cpu()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r06_dropin_budget.jsonl"), "a") as f:
    for r in rows:
        f.write(json.dumps(r) + "\n")
