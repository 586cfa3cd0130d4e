"""The OPT-IN reduced-precision forward kernels (bf16 matrix pipe; BASELINE.json configs[4] "bf16 vs fp32 tolerance sweep"):
checked against the numpy emulation of the same operand splitting (oracle/lowp.py) and against the fp32 HIP path.
These kernels are NOT within the 1e-5 parity bar (bf16x6 excepted, nearly) -- the numbers asserted here are the sweep.
Needs an MI355X."""
import numpy as np
import pytest
import torch

from conftest import load_golden, tape

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from bnn_chaos_model_amd import ops as o
    return o


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


FORMS = {"bf16": (1, "bf16"), "bf16x3": (2, "bf16"), "bf16x6": (3, "bf16"), "f16": (1, "f16"), "f16x3": (2, "f16")}


@pytest.mark.parametrize("prec,tol", (("bf16", 2e-4), ("bf16x3", 2e-5), ("bf16x6", 2e-5), ("f16", 1e-4), ("f16x3", 2e-5)))
def test_pooled_summary_matches_the_emulation(ops, inputs, prec, tol):
    """With the two pool-noise draws set to zero the kernel's summary is [time mean | sqrt(var + 1e-5)] of the latents: compare
    with the float64 emulation of the same operand splitting.  What is left is fp32 accumulation inside the matrix pipe and in
    the pool (Welford), orders of magnitude below the effect of the splitting itself for bf16."""
    from oracle import lowp
    z = load_golden("case_swagfast_v50_0_slow.npz")
    x = inputs["slow"]
    B = x.shape[0]
    eps = np.zeros((1, B, 2, 20), np.float32)
    out, pre, summ = ops.forward(dev(x), dev(z["w"][None]), eps=dev(eps), debug=True, precision=prec)
    ns, fmt = FORMS[prec]
    want = lowp.pooled_summary(lowp.feature_nn(x, z["w"], ns, fmt))
    got = summ.cpu().numpy()[0].astype(np.float64)
    scale = np.abs(want).max(1, keepdims=True)
    assert (np.abs(got - want) <= tol * scale).all(), (np.abs(got - want) / scale).max()
    # ragged batch sizes run the same code: a 5-system batch equals the first 5 rows
    out5 = ops.forward(dev(x[:5]), dev(z["w"][None]), eps=dev(eps[:, :5]), precision=prec)
    assert torch.equal(out5, out[:, :5])


def test_precision_sweep_against_fp32_and_reference(ops, inputs, swag_states, capsys):
    """The sweep itself on the reference's fixture (v50_0, 'slow' inputs, the reference's own normals): |d mu|, |d std| of each
    reduced-precision form against the fp32 HIP path and against the reference's outputs."""
    z = load_golden("case_swagfast_v50_0_slow.npz")
    tp = tape(z)
    x = inputs["slow"]
    eps = np.stack([tp[2][1], tp[3][1]], axis=1)[None]
    f32 = ops.forward(dev(x), dev(z["w"][None]), eps=dev(eps)).cpu().numpy()[0].astype(np.float64)
    rows = {}
    for prec in FORMS:
        o = ops.forward(dev(x), dev(z["w"][None]), eps=dev(eps), precision=prec).cpu().numpy()[0].astype(np.float64)
        d = np.abs(o - f32)
        rel_ref = (np.abs(o - z["out"]) / np.abs(z["out"])).max()
        rows[prec] = (d[:, 0].max(), d[:, 1].max(), np.median(d[:, 0]), rel_ref)
    with capsys.disabled():
        for k, (a, b, m, rr) in rows.items():
            print(f"\n  lowp sweep [{k:7s}] max |d mu| = {a:.3e}  max |d std| = {b:.3e}  median |d mu| = {m:.3e}  max rel vs reference = {rr:.3e}", end="")
    assert 1e-4 < rows["bf16"][0] < 1.0            # plain bf16: three to four orders of magnitude above the parity bar
    assert rows["bf16x3"][0] < rows["bf16"][0] / 20  # 16 significant bits
    assert rows["bf16x6"][0] < 2e-5 and rows["bf16x6"][3] < 1e-5   # 24 bits: fp32-level error; within the bar on this fixture
    assert rows["f16"][0] < rows["bf16"][0] / 3                    # 11 significant bits against 8
    assert rows["f16x3"][0] < rows["bf16x3"][0] / 8 and rows["f16x3"][0] < 5e-5   # 22 bits: close to fp32 at the cost of bf16x3


def test_lowp_is_invariant_to_sharding_and_refuses_what_it_does_not_build(ops, swag_states):
    import bench
    wa = dev(np.stack([swag_states[0]["w_avg"], swag_states[12]["w_avg"]]))
    w2 = dev(np.stack([swag_states[0]["w2_avg"], swag_states[12]["w2_avg"]]))
    pd = dev(np.stack([swag_states[0]["pre_D"], swag_states[12]["pre_D"]]))
    x = bench.synthetic_x(700, torch.device("cuda"), 5)
    idx = torch.as_tensor((np.arange(20) % 2).astype(np.int32))
    for prec in FORMS:
        a = ops.multiswag(x, wa, w2, pd, idx, nchunks=10, philox_seed=3, system_id0=1000, precision=prec)
        assert a.shape == (2, 700, 2) and torch.isfinite(a).all()
        full = ops.multiswag(x, wa, w2, pd, idx, philox_seed=3, system_id0=1000, precision=prec)
        part = ops.multiswag(x[333:].contiguous(), wa, w2, pd, idx, philox_seed=3, system_id0=1333, precision=prec)
        assert torch.equal(part, full[:, 333:])
        f32 = ops.multiswag(x, wa, w2, pd, idx, philox_seed=3, system_id0=1000)
        assert (full - f32).abs().max() < (1.0 if prec in ("bf16", "f16") else 0.05)
    W = ops.swag_draw(wa, w2, pd, idx, philox_seed=3)
    with pytest.raises(NotImplementedError):
        ops.forward(x, W, noisy=True, precision="bf16")
    with pytest.raises(ValueError):
        ops.forward(x, W, precision="fp8")
    from bnn_chaos_model_amd import _native as N
    with pytest.raises(N.NativeError):   # another column mask: not built
        ops.forward(x, W, precision="bf16", plan=ops.get_plan(zero_mask=1 << 7))


@pytest.mark.parametrize("T", (8, 36))
def test_other_series_lengths(ops, swag_states, T):
    """T is a run-time size (a multiple of 4): the reduced-precision kernels agree with their emulation for short series too, and
    with the fp32 path at fp32 level for the six-product form."""
    from oracle import lowp
    rng = np.random.default_rng(T)
    B = 23
    x = (rng.standard_normal((B, 1, 41)) + 0.1 * rng.standard_normal((B, T, 41))).astype(np.float32)
    w = swag_states[12]["w_avg"]
    eps = np.zeros((1, B, 2, 20), np.float32)
    f32, _, s32 = ops.forward(dev(x), dev(w[None]), eps=dev(eps), debug=True)
    for prec, tol in (("bf16", 3e-4), ("f16x3", 2e-5), ("bf16x6", 2e-5)):
        out, _, summ = ops.forward(dev(x), dev(w[None]), eps=dev(eps), debug=True, precision=prec)
        ns, fmt = FORMS[prec]
        lat = lowp.feature_nn(x, w, ns, fmt)
        want = lowp.pooled_summary(lat)
        got = summ.cpu().numpy()[0].astype(np.float64)
        scale = np.abs(want).max(1, keepdims=True)
        assert (np.abs(got - want) <= tol * scale).all(), (prec, (np.abs(got - want) / scale).max())
    assert (out - f32).abs().max() < 1e-4 and (summ - s32).abs().max() < 1e-4 * s32.abs().max()


def test_surface_passes_precision_through(swag_states, tmp_path, inputs):
    """FeatureRegressor.sample_full_swag_many(precision=...) is ops.multiswag(precision=...): f16x3 stays at fp32 level on the
    5-planet loop shape, with the reference's RNG consumption."""
    import json
    from bnn_chaos_model_amd import checkpoint
    from bnn_chaos_model_amd.regression import FeatureRegressor
    for i in (0, 12):
        z = load_golden(f"swag_v50_{i}.npz")
        checkpoint.write_swag_file(str(tmp_path / f"s_v50_{i:02d}_output.pkl"), json.loads(str(z["hparams_json"])), json.loads(str(z["swa_params_json"])),
                                   torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"]))
    fr = FeatureRegressor(cuda=False, filebase=str(tmp_path / "*v50*output.pkl"), sort=True)
    x = torch.tensor(inputs["slow"][:30])
    outs = {}
    for prec in ("f32", "f16x3", "bf16"):
        np.random.seed(1); torch.manual_seed(1)
        outs[prec] = fr.sample_full_swag_many(x, samples=4, chunks=10, precision=prec)
    assert (outs["f16x3"] - outs["f32"]).abs().max() < 2e-5
    assert 1e-5 < (outs["bf16"] - outs["f32"]).abs().max() < 1.0


def test_half_forms_saturate_out_of_range_inputs(ops, inputs):
    """IEEE half has no exponent headroom: the scripts' constant-4 fill of unstable systems standardises the mass columns to
    1.9e5 > 65 504.  The half forms then saturate (MODE.FP16_OVFL: finite, but wrong -- those rows' outputs are discarded by the
    script); the bfloat16 forms keep fp32's range and stay accurate on the same input."""
    z = load_golden("case_swagfast_v50_0_const4.npz")
    tp = tape(z)
    x = inputs["const4"]
    assert np.abs(x).max() > 65504
    eps = np.stack([tp[2][1], tp[3][1]], axis=1)[None]
    for prec in ("f16", "f16x3"):
        o = ops.forward(dev(x), dev(z["w"][None]), eps=dev(eps), precision=prec)
        assert torch.isfinite(o).all()
    o = ops.forward(dev(x), dev(z["w"][None]), eps=dev(eps), precision="bf16x6").cpu().numpy()[0]
    assert (np.abs(o - z["out"]) <= 1e-5 * np.abs(z["out"])).all()
    # the precondition is checkable: half_range_exceeded names the rows, the 'slow' inputs have none
    bad = ops.half_range_exceeded(dev(x))
    assert bad.shape == (x.shape[0],) and bool(bad.all())
    assert not bool(ops.half_range_exceeded(dev(inputs["slow"])).any())


def test_surface_warns_when_half_cannot_hold_the_inputs(swag_states, tmp_path, inputs):
    import json
    import warnings
    from bnn_chaos_model_amd import checkpoint
    from bnn_chaos_model_amd.regression import FeatureRegressor
    z = load_golden("swag_v50_0.npz")
    checkpoint.write_swag_file(str(tmp_path / "s_v50_00_output.pkl"), json.loads(str(z["hparams_json"])), json.loads(str(z["swa_params_json"])),
                               torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"]))
    fr = FeatureRegressor(cuda=False, filebase=str(tmp_path / "*v50*output.pkl"), sort=True)
    x = torch.tensor(np.concatenate([inputs["slow"][:4], inputs["const4"][:2]]))
    with pytest.warns(RuntimeWarning, match="2 of 6 rows"):
        fr.sample_full_swag_many(x, samples=2, rng="philox", precision="f16x3")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        fr.sample_full_swag_many(x, samples=2, rng="philox", precision="bf16x6")   # fp32 range: nothing to warn about
        fr.sample_full_swag_many(x[:4], samples=2, rng="philox", precision="f16x3")
