import json, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bnn_chaos_model_amd import ops
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
N = 375000
g = torch.Generator(device="cuda").manual_seed(1)
z = np.load("tests/golden/ensemble_v50.npz")
wa, w2, pd = (torch.as_tensor(z[k]).cuda() for k in ("w_avg", "w2_avg", "pre_D"))
x = torch.randn(N, 100, 41, generator=g, device="cuda")
idx100 = (torch.arange(100, dtype=torch.int32) % 30).cuda()
out = torch.empty(100, N, 2, device="cuda")
withscan = timed(lambda: ops.multiswag(x, wa, w2, pd, idx100, philox_seed=3, out=out))
for frac in (1000, 100):
    hurt = torch.randperm(N, generator=torch.Generator().manual_seed(5))[: N // frac].cuda()
    xe = x.clone(); xe[hurt, 7, 12] = float("inf")
    t_e = timed(lambda: ops.multiswag(xe, wa, w2, pd, idx100, philox_seed=3, out=out))
    n = 100 * int(hurt.numel())
    print(json.dumps({"listed": int(hurt.numel()), "items": n, "ms_extra_exact": round(t_e - withscan, 3), "items_per_s": round(n / (t_e - withscan) * 1e3)}))
    xs = xe[hurt[:64]].contiguous()
    t_f = timed(lambda: ops.multiswag(xs, wa, w2, pd, idx100[:10], philox_seed=3, single_launch=True))
    t_f0 = timed(lambda: ops.multiswag(xs, wa, w2, pd, idx100[:10], philox_seed=3, single_launch=True, assume_finite=True))
    print(json.dumps({"fused small call, 64 listed systems x 10 draws": round(t_f - t_f0, 3)}))
