"""hparams['fix_megno'] = True on the GPU (reference spock_reg_model.py:360-362, 480-491, 509-510): 42-wide summary (the last two
entries are the time mean / unbiased std of the RAW MEGNO column), d = 7665.  Dead for every pretrained checkpoint; checked against
a fixture the unmodified reference produced with the flag set (tests/golden/make_golden_megno.py) and against the oracle bit for
bit where the arithmetic is pinned.  Needs an MI355X."""
import json

import numpy as np
import pytest
import torch

from conftest import close_report, load_golden

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def tp(z, pfx):
    return [z[f"{pfx}_{i:03d}"] for i in range(int(z[pfx + "_n"]))]


@pytest.fixture(scope="module")
def z():
    return load_golden("case_megno.npz")


@pytest.fixture(scope="module")
def ops():
    from bnn_chaos_model_amd import ops as o
    return o


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def megno_mask(ops):
    return ops.zero_mask_from_flags(fix_megno=True, fix_megno2=False)


def test_forward_quiet_and_noisy_vs_reference_and_oracle(ops, orc, z):
    plan = ops.get_plan(megno_mask(ops), 0.5, fix_megno=True)
    assert plan.d == 7665 and plan.summary_width == 42
    arch = orc.make_arch(T=100, fix_megno=True)
    x, W = dev(z["x"]), dev(z["swagfast_w"][None])
    for noisy in (0, 1):
        t = tp(z, f"forward_noisy{noisy}_tape")
        e1, e2 = (t[1], t[2]) if noisy else (t[0], t[1])
        eps = dev(np.stack([e1, e2], 1)[None])
        kw = dict(eps_in=dev(t[0][None]), eps_sum=dev(t[3][None])) if noisy else {}
        out, pre, summ = ops.forward(x, W, eps=eps, plan=plan, debug=True, **kw)
        nbad, mx = close_report(out[0].cpu().numpy(), z[f"forward_noisy{noisy}_out"])          # the reference itself, 1e-5 relative
        assert nbad == 0, (noisy, nbad, mx)
        assert summ.shape == (1, 16, 42)
        # the oracle on the kernel's accumulation schedule: summary (all 42 entries) and pre-clamp outputs bit for bit
        sched = orc.make_schedule([plan.layer_order(l, noisy=bool(noisy)) for l in range(6)], pool_parts=4)
        okw = dict(eps_in=t[0], eps_sum=t[3]) if noisy else {}
        o, ex = orc.forward(z["x"], z["swagfast_w"], e1, e2, arch=arch, sched=sched, extras=True, **okw)
        if noisy:   # expf of the noise scales differs by an ulp between libm and the device
            nbad, mx = close_report(summ[0].cpu().numpy(), ex["summary"], rtol=2e-6, atol=2e-6)
            assert nbad == 0, (nbad, mx)
        else:
            assert np.array_equal(summ[0].cpu().numpy(), ex["summary"])
            assert np.array_equal(pre[0].cpu().numpy(), ex["pre_clamp"])
        assert np.abs(out[0].cpu().numpy() - o).max() <= 2e-6
        ref = z[f"forward_noisy{noisy}_summary"]
        assert np.abs(summ[0, :, 40:].cpu().numpy() - ref[:, 40:]).max() <= 1e-5 * np.abs(ref[:, 40:]).max()


def test_multiswag_and_draw_vs_reference(ops, orc, z):
    """forward_swag_fast (:878-908): draw (d = 7665, one negative-variance element) + forward, both launch modes, vs the reference's
    outputs and sampled weights; the draw equals the oracle's bit for bit."""
    plan = ops.get_plan(megno_mask(ops), 0.5, fix_megno=True)
    t = tp(z, "swagfast_tape")
    wa, w2, pd = dev(z["w_avg"][None]), dev(z["w2_avg"][None]), dev(z["pre_D"][None])
    idx = torch.zeros(1, dtype=torch.int32)
    z1, z2 = dev(t[0]), dev(t[1].reshape(1, -1))
    W = ops.swag_draw(wa, w2, pd, idx, z1, z2, scale=0.5, plan=plan)
    assert W.shape == (1, 7665)
    assert np.array_equal(W[0].cpu().numpy(), orc.swag_draw(z["w_avg"], z["w2_avg"], z["pre_D"], t[0], t[1]))
    assert np.abs(W[0].cpu().numpy() - z["swagfast_w"]).max() <= 2e-6
    eps = dev(np.stack([t[2], t[3]], 1)[None])
    outs = [ops.multiswag(dev(z["x"]), wa, w2, pd, idx, z1, z2, eps, plan=plan, single_launch=sl) for sl in (True, False)]
    assert torch.equal(outs[0], outs[1])
    nbad, mx = close_report(outs[0][0].cpu().numpy(), z["swagfast_out"])
    assert nbad == 0, (nbad, mx)
    # a mask other than the v50 one: the 41-column form (here: only MEGNO zeroed) against the oracle
    m2 = 1 << 7
    plan2 = ops.get_plan(m2, 0.5, fix_megno=True)
    o2 = ops.multiswag(dev(z["x"]), wa, w2, pd, idx, z1, z2, eps, plan=plan2)[0].cpu().numpy()
    sched = orc.make_schedule([plan2.layer_order(l) for l in range(6)], pool_parts=4)
    want = orc.forward(z["x"], W[0].cpu().numpy(), t[2], t[3], arch=orc.make_arch(T=100, zero_mask=m2, fix_megno=True), sched=sched)
    assert np.abs(o2 - want).max() <= 2e-6


def test_in_kernel_noise_equals_explicit_and_regress_is_the_tail(ops, z):
    plan = ops.get_plan(megno_mask(ops), 0.5, fix_megno=True)
    x = dev(np.tile(z["x"], (3, 1, 1))[:37])
    W = dev(np.stack([z["swagfast_w"], z["w_avg"], z["swagfast_w"]]))
    B, R, seed = 37, 3, 91
    a = ops.forward(x, W, philox_seed=seed, draw_id0=4, system_id0=1000, noisy=True, plan=plan)
    eps = ops.philox_normal(2, seed, 4, R, B=B, system_id0=1000)
    e_in = ops.philox_normal(3, seed, 4, R, width=100, B=B, system_id0=1000)
    e_sum = ops.philox_normal(4, seed, 4, R, width=42, B=B, system_id0=1000)
    assert e_sum.shape == (R, B, 42)
    b, pre, summ = ops.forward(x, W, eps=eps, eps_in=e_in, eps_sum=e_sum, plan=plan, debug=True)
    assert torch.equal(a, b)
    e40 = ops.philox_normal(4, seed, 4, R, B=B, system_id0=1000)
    assert torch.equal(e40, e_sum[..., :40])                        # the first 40 entries are the 40-wide stream
    quiet, pre_q, summ_q = ops.forward(x, W, philox_seed=seed, draw_id0=4, system_id0=1000, plan=plan, debug=True)
    assert summ_q.shape == (R, B, 42)
    out_r, pre_r = ops.regress(summ_q, W, plan=plan, debug=True)   # predict_instability on the explicit 42-wide summary
    assert torch.equal(pre_r, pre_q) and torch.equal(out_r, quiet)
    with pytest.raises(Exception):
        ops.forward(x, W, plan=plan, precision="bf16")              # not built for fix_megno
    with pytest.raises(ValueError):
        ops.forward(x, W[:, :7583].contiguous(), plan=plan)


def test_surface_replays_the_reference_seeds(z, tmp_path):
    """load_swag on a checkpoint whose hparams say fix_megno=True -> the module API with the reference's RNG order reproduces the
    reference run seed for seed (forward_swag_fast :878-908, forward :486-528, compute_summary_stats / predict_instability)."""
    from bnn_chaos_model_amd import checkpoint, spock_reg_model as srm
    hp = json.loads(str(z["hparams_json"]))
    for k, v in list(hp.items()):
        if isinstance(v, str) and v in ("True", "False"):
            hp[k] = v == "True"
    p = tmp_path / "megno_output.pkl"
    checkpoint.write_swag_file(str(p), hp, json.loads(str(z["swa_params_json"])), torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]),
                               torch.tensor(z["pre_D"]))
    m = srm.load_swag(str(p)).eval()
    assert m.fix_megno and m.flatten().numel() == 7665 and m.state_dict()["regress_nn.0.weight"].shape == (40, 42)
    x = torch.tensor(z["x"])
    torch.manual_seed(5150)
    out = m.forward_swag_fast(x, scale=0.5)
    nbad, mx = close_report(out.numpy(), z["swagfast_out"])
    assert nbad == 0, (nbad, mx)
    assert np.abs(m.flatten().numpy() - z["swagfast_w"]).max() <= 2e-6        # the sampled weights stay loaded (:838)
    for noisy in (False, True):
        torch.manual_seed(5151 + int(noisy))
        o = m(x, noisy_val=noisy)
        nbad, mx = close_report(o.numpy(), z[f"forward_noisy{int(noisy)}_out"])
        assert nbad == 0, (noisy, nbad, mx)
    # compute_summary_stats is 40 wide and unmasked (:416-435); predict_instability takes the 42-wide summary (:437-442)
    torch.manual_seed(5151)
    s40 = m.compute_summary_stats(m._masked(x))
    assert s40.shape == (16, 40)
    nbad, mx = close_report(s40.numpy(), z["forward_noisy0_summary"][:, :40], rtol=2e-5, atol=2e-5)
    assert nbad == 0, (nbad, mx)
    mu, std = m.predict_instability(torch.tensor(z["forward_noisy0_summary"]))
    nbad, mx = close_report(torch.cat((mu, std), 1).numpy(), z["forward_noisy0_out"])
    assert nbad == 0, (nbad, mx)
    # sample(): the estimator runs (42-wide summary noise drawn from torch's generator in the reference's order)
    np.random.seed(0); torch.manual_seed(0)
    s = m.sample(x, samples=4)
    assert s.shape == (16,) and np.isfinite(s).all() and (s > 0).all()
