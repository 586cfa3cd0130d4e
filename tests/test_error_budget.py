"""Error budget against the float64 truth AT SCALE (round 4): 10^6 (system, draw) evaluations of the configs[4] shapes (100 000 rows,
10 chunks, 10 samples, a random ensemble member per chunk per sample: figures/multiswag_5_planet.py:287, 295-298) through the fp32 HIP
path, the two "fp32-level" reduced-precision forms (bf16x6, f16x3) and the 16-bit split (bf16x3), and through the oracle in float64
(orc64_*: the same op sequence, every weight draw and every pool normal the kernel's own Philox numbers).  Per arithmetic: the largest
and the 99.9th-percentile relative error of (mu, std) and the number of evaluations beyond 1e-5 |truth| -- the tracked table is
profiles/r04_error_budget.json (this test writes gpurun_out/r4_error_budget.json on the GPU box).

What the numbers decide: SURVEY.md section 6 measured 2e-5 as the largest distance of the REFERENCE's fp32 result from the fp64 truth
(4 000 systems).  A form whose distribution of errors against the truth is no wider than the fp32 path's may be documented as "within
the reference's fp32 noise floor"; the others are approximate modes.  Needs an MI355X (and 16 host threads for the float64 oracle)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


def test_error_budget_against_float64_truth_at_scale():
    from bnn_chaos_model_amd import ops
    from oracle import oracle as orc
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
    B, NCH, SAMPLES, SEED = 100_000, 10, 10, 424242
    J = NCH * SAMPLES
    z = load_golden("ensemble_v50.npz")
    wa, w2, pd = (torch.as_tensor(z[k]).cuda() for k in ("w_avg", "w2_avg", "pre_D"))
    S, d, K = pd.shape
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(B, 100, 41, generator=g, device="cuda") * 0.1 + torch.randn(B, 1, 41, generator=g, device="cuda")   # SURVEY 8d "slow" inputs
    x[:, :, 0] = torch.linspace(-1.71, 1.74, 100, device="cuda")[None]
    seed_idx = torch.as_tensor(np.random.default_rng(3).integers(0, S, J).astype(np.int32)).cuda()
    plan = ops.get_plan()
    W = ops.swag_draw(wa, w2, pd, seed_idx, philox_seed=SEED, draw_id0=0, plan=plan)
    outs = {"f32": ops.forward(x, W, nchunks=NCH, philox_seed=SEED, draw_id0=0, system_id0=0, plan=plan)}
    for prec in ("bf16x6", "f16x3", "bf16x3"):
        outs[prec] = ops.forward(x, W, nchunks=NCH, philox_seed=SEED, draw_id0=0, system_id0=0, plan=plan, precision=prec)
    # the truth: float64 oracle, draw included, on the kernel's own normals
    z1 = ops.philox_normal(0, SEED, 0, J, width=d).cpu().numpy()
    z2 = ops.philox_normal(1, SEED, 0, J, width=K).cpu().numpy()
    eps = ops.philox_normal(2, SEED, 0, SAMPLES, B=B, system_id0=0).cpu().numpy()
    truth = orc.multiswag(x.cpu().numpy(), z["w_avg"], z["w2_avg"], z["pre_D"], seed_idx.cpu().numpy(), z1, z2, eps, nchunks=NCH, dtype=np.float64)
    assert truth.shape == (SAMPLES, B, 2) and np.isfinite(truth).all()
    table = {"evals": int(SAMPLES * B), "shape": f"{B} rows x {NCH} chunks x {SAMPLES} samples, 30 pretrained members, in-kernel Philox normals",
             "bar": "1e-5 * |truth| per output (mu and std counted separately)", "arithmetics": {}}
    for name, o in outs.items():
        rel = np.abs(o.cpu().numpy().astype(np.float64) - truth) / np.abs(truth)
        table["arithmetics"][name] = {
            "max_rel": float(rel.max()), "p99_9_rel": float(np.quantile(rel, 0.999)), "p99_rel": float(np.quantile(rel, 0.99)),
            "median_rel": float(np.median(rel)), "exceed_1e-5": int((rel > 1e-5).sum()), "exceed_frac": float((rel > 1e-5).mean()),
            "max_abs_mu": float(np.abs(o.cpu().numpy()[..., 0] - truth[..., 0]).max()), "max_abs_std": float(np.abs(o.cpu().numpy()[..., 1] - truth[..., 1]).max())}
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "r4_error_budget.json"), "w") as f:
            json.dump(table, f, indent=1)
    except OSError:
        pass
    print("\n" + json.dumps(table["arithmetics"], indent=1))
    a = table["arithmetics"]
    # the fp32 parity path sits at the fp32 noise floor of the algorithm: nothing beyond a few 1e-5, all but ~1e-3 of the outputs within 1e-5
    assert a["f32"]["max_rel"] < 1e-4 and a["f32"]["p99_9_rel"] < 2e-5 and a["f32"]["exceed_frac"] < 5e-3
    # bf16x6 (24 significant bits) and f16x3 (22 bits): the same order of magnitude as the fp32 path itself
    for name in ("bf16x6", "f16x3"):
        assert a[name]["max_rel"] < 3e-4 and a[name]["p99_9_rel"] < 5e-5, (name, a[name])
    # bf16x3 (16 bits) is an approximate mode: two orders of magnitude wider
    assert a["bf16x3"]["p99_9_rel"] > a["f32"]["p99_9_rel"] * 5
