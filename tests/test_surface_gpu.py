"""The reference's Python surface (spock_reg_model / FeatureRegressor) on top of the HIP kernels, driven exactly
as the evaluation scripts drive it, against outputs captured from the unmodified reference.  Needs an MI355X."""
import json

import numpy as np
import pytest
import torch

from conftest import close_report, load_golden, tape

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ckpt_dir(tmp_path_factory):
    """The two converted pretrained seeds written back as reference-format checkpoints (names contain 'v50')."""
    from bnn_chaos_model_amd import checkpoint
    d = tmp_path_factory.mktemp("pretrained")
    for i in (0, 12):
        z = load_golden(f"swag_v50_{i}.npz")
        checkpoint.write_swag_file(str(d / f"steps=300000_v50_{i:02d}_output.pkl"), json.loads(str(z["hparams_json"])),
                                   json.loads(str(z["swa_params_json"])), torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]),
                                   torch.tensor(z["pre_D"]))
    return d


@pytest.fixture(scope="module")
def models(ckpt_dir):
    from bnn_chaos_model_amd import spock_reg_model as srm
    return {i: srm.load_swag(str(ckpt_dir / f"steps=300000_v50_{i:02d}_output.pkl")).cpu().eval() for i in (0, 12)}


@pytest.mark.parametrize("si", (0, 12))
@pytest.mark.parametrize("xname", ("slow", "iid", "const4"))
def test_forward_swag_fast_replays_reference_seed(si, xname, models, inputs):
    """torch.manual_seed(s); model.forward_swag_fast(x, 0.5) -- same call, same seed, same numbers as the reference."""
    z = load_golden(f"case_swagfast_v50_{si}_{xname}.npz")
    m = models[si]
    x = torch.tensor(inputs[xname])
    torch.manual_seed(int(z["torch_seed"]))
    out = m.forward_swag_fast(x, scale=0.5)
    assert out.device == x.device and out.shape == (x.shape[0], 2) and out.dtype == torch.float32
    nbad, mx = close_report(out.numpy(), z["out"])
    assert nbad == 0, (nbad, mx)
    # the sampled weights are left loaded in the module, as in the reference (:838)
    assert np.abs(m.flatten().numpy().astype(np.float64) - z["w"]).max() <= 2e-6
    torch.manual_seed(int(z["torch_seed"]))
    out2 = m.forward_swag(x, scale=0.5)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("si", (0, 12))
@pytest.mark.parametrize("noisy", (False, True))
def test_varmodel_forward_replays_reference_seed(si, noisy, models, inputs):
    z = load_golden(f"case_forward_v50_{si}_noisy{int(noisy)}.npz")
    m = models[si]
    m.load(torch.tensor(z["w"]))
    torch.manual_seed(int(z["torch_seed"]))
    out = m(torch.tensor(inputs["slow"]), noisy_val=noisy)
    nbad, mx = close_report(out.numpy(), z["out"])
    assert nbad == 0, (nbad, mx)


def test_varmodel_sample_replays_reference_seed(models, inputs):
    """VarModel.sample (:530-545) = mean over `samples` noisy forwards of mu + n * std (n = np.random.randn).  The 1e-5 relative
    bar holds for mu and std of every forward; through the estimator it becomes 1e-5 * mean_s(|mu_s| + |n_s| * std_s), which is
    the tolerance used here (n_s replayed from the same seed), with zero exceedances."""
    z = load_golden("case_sample_v50_0.npz")
    m = models[0]
    m.load(torch.tensor(z["w"]))
    samples = int(z["samples"])
    x = torch.tensor(inputs["slow"])
    torch.manual_seed(3000)
    np.random.seed(3000)
    s = m.sample(x, samples=samples)
    assert isinstance(s, np.ndarray) and s.dtype == np.float64 and s.shape == (32,)
    # the estimator's terms, by replaying the same generator streams through forward()
    torch.manual_seed(3000)
    np.random.seed(3000)
    m.cpu()
    scale = np.zeros(32)
    acc = []
    for _ in range(samples):
        o = m(x).detach().numpy()
        n = np.random.randn(32)
        acc.append(o[:, 0] + n * o[:, 1])
        scale += np.abs(o[:, 0]) + np.abs(n) * o[:, 1]
    assert np.array_equal(np.average(acc, axis=0), s)
    err = np.abs(s - z["out"])
    tol = 1e-5 * scale / samples
    assert (err <= tol).all(), (err / tol).max()


def test_sample_weights_flatten_load(models):
    z = load_golden("case_swagfast_v50_0_slow.npz")
    m = models[0]
    torch.manual_seed(int(z["torch_seed"]))
    m.sample_weights(scale=0.5)
    assert np.abs(m.flatten().numpy().astype(np.float64) - z["w"]).max() <= 2e-6
    m.sample_weights(scale=0)
    assert torch.equal(m.flatten(), m.w_avg)  # scale = 0 => w == w_avg


def test_compute_summary_stats_and_predict_instability(models, inputs):
    """predict_instability on the REFERENCE's summary meets the 1e-5 relative bar.  compute_summary_stats is an intermediate:
    each entry is a sum of two terms (eps * std_in_mu + sample_mu, :428-431) that may cancel, so the bar is applied to the
    terms' scale: |a - b| <= 1e-5 * (|b| + S) with S = the largest |entry| of the same kind (mean-like / std-like) in the row."""
    z = load_golden("case_swagfast_v50_0_slow.npz")
    m = models[0]
    m.load(torch.tensor(z["w"]))
    x = torch.tensor(inputs["slow"])
    xm = x.clone()
    xm[..., [1, 2, 3, 4, 5, 6, 7, 38, 39, 40]] = 0
    torch.manual_seed(int(z["torch_seed"]))
    torch.randn((1, 7583)); torch.randn((30, 1))  # skip the weight-draw part of the reference's stream
    summ = m.compute_summary_stats(xm).numpy().astype(np.float64)
    ref = z["summary"].astype(np.float64)
    S = np.concatenate([np.abs(ref[:, :20]).max(1, keepdims=True).repeat(20, 1), np.abs(ref[:, 20:]).max(1, keepdims=True).repeat(20, 1)], 1)
    err = np.abs(summ - ref)
    assert (err <= 1e-5 * (np.abs(ref) + S)).all(), (err / (1e-5 * (np.abs(ref) + S))).max()
    mu, std = m.predict_instability(torch.tensor(z["summary"]))
    assert mu.shape == (32, 1) and std.shape == (32, 1)
    nbad, mx = close_report(torch.cat((mu, std), 1).numpy(), z["out"])
    assert nbad == 0, (nbad, mx)


def test_noise_helpers_follow_the_reference_stream(models, inputs):
    """add_input_noise / add_summary_noise (:444-450) draw one randn_like each and scale by exp(logvar / 2)."""
    m = models[0]
    x = torch.tensor(inputs["slow"][:4])
    torch.manual_seed(5)
    got = m.add_input_noise(x)
    torch.manual_seed(5)
    want = x + torch.randn_like(x) * torch.exp(m.flatten()[:41].cpu() / 2)
    assert torch.equal(got, want)
    s = torch.randn(4, 40)
    torch.manual_seed(6)
    got = m.add_summary_noise(s)
    torch.manual_seed(6)
    want = s + torch.randn_like(s) * torch.exp(m.flatten()[41:81].cpu() / 2)
    assert torch.equal(got, want)


def test_feature_regressor_sample_full_swag(ckpt_dir, inputs):
    """FeatureRegressor.sample_full_swag x3 (figures/spock/regression.py:74-92) == the captured reference run."""
    from bnn_chaos_model_amd.regression import FeatureRegressor
    z = load_golden("case_multiswag_grid.npz")
    fr = FeatureRegressor(cuda=False, filebase=str(ckpt_dir / "*v50*output.pkl"), sort=True)
    assert len(fr.swag_ensemble) == 2 and fr.ssX.mean_.shape == (41,) and fr.cuda is False
    x = torch.tensor(inputs["slow"])
    np.random.seed(4000)
    torch.manual_seed(4000)
    outs = torch.cat([fr.sample_full_swag(x)[None].detach() for _ in range(3)])
    nbad, mx = close_report(outs.numpy(), z["out"])
    assert nbad == 0, (nbad, mx)
    # the batched driver consumes the generators identically and gives the same numbers in one launch
    np.random.seed(4000)
    torch.manual_seed(4000)
    many = fr.sample_full_swag_many(x, samples=3, chunks=1)
    assert torch.equal(many, outs)
    with pytest.raises(NotImplementedError):
        fr.sample_full_swag(torch.zeros(2, 100, 40))


def test_five_planet_mc_loop(ckpt_dir, inputs):
    """figures/multiswag_5_planet.py:295-298, literally, and as one launch."""
    from bnn_chaos_model_amd.regression import FeatureRegressor
    z = load_golden("case_chunk_loop.npz")
    model = FeatureRegressor(cuda=False, filebase=str(ckpt_dir / "*v50*output.pkl"), sort=True)
    Xflat = torch.tensor(inputs["slow"][:30])
    samples = int(z["samples"])
    np.random.seed(5000)
    torch.manual_seed(5000)
    time = torch.cat([
        torch.cat([model.sample_full_swag(Xpart).detach().cpu() for Xpart in torch.chunk(Xflat, chunks=10)])[None]
        for _ in range(samples)], dim=0)
    nbad, mx = close_report(time.numpy(), z["out"])
    assert nbad == 0, (nbad, mx)
    np.random.seed(5000)
    torch.manual_seed(5000)
    one = model.sample_full_swag_many(Xflat, samples=samples, chunks=10)
    assert torch.equal(one, time)


def test_philox_mode_statistics_match_torch_mode(ckpt_dir, inputs):
    """rng='philox' draws different numbers from the same distributions: predictive moments agree statistically."""
    from bnn_chaos_model_amd.regression import FeatureRegressor
    fr = FeatureRegressor(cuda=False, filebase=str(ckpt_dir / "*v50*output.pkl"), sort=True)
    x = torch.tensor(inputs["slow"])
    np.random.seed(1); torch.manual_seed(1)
    a = fr.sample_full_swag_many(x, samples=400, chunks=1, rng="torch")
    np.random.seed(1)
    b = fr.sample_full_swag_many(x, samples=400, chunks=1, rng="philox", philox_seed=7)
    assert a.shape == b.shape == (400, 32, 2)
    ma, mb = a[..., 0].mean(0), b[..., 0].mean(0)
    sa = a[..., 0].std(0) + 1e-3
    assert ((ma - mb).abs() / sa * (400 ** 0.5) < 6).all()  # means within 6 standard errors, system by system


def test_cuda_inputs_stay_on_gpu(models, inputs):
    m = models[0]
    x = torch.tensor(inputs["slow"]).cuda()
    m.cuda()
    try:
        out = m.forward_swag_fast(x)
        assert out.is_cuda and out.shape == (32, 2) and torch.isfinite(out).all()
        m.rng = "philox"
        m.philox_seed = 11
        o1 = m.forward_swag_fast(x)
        assert o1.is_cuda and torch.isfinite(o1).all() and torch.isfinite(m.flatten()).all()
    finally:
        m.rng = "torch"
        m.cpu()


def test_gpu_resident_route_draws_the_reference_calls_in_order(models, inputs, swag_states):
    """FeatureRegressor(cuda=True)'s per-call route: model, x and generator on the GPU.  forward_swag_fast draws into buffers the model keeps
    per (stream, batch size) -- randn(out=), normal_() on strided views -- and must consume the GPU generator exactly like the reference's four
    calls in its order (spock_reg_model.py:830-831 randn((1,d)), randn((K,1)); :426-427 two randn_like([B, latent])): same outputs as
    ops.multiswag fed with those four torch.randn results, call after call (buffer reuse), for another batch size, and on another stream;
    the weights the module then "has loaded" (:838) are those of ITS last call."""
    from bnn_chaos_model_amd import ops
    m = models[0]
    st = swag_states[0]
    g = torch.device("cuda", torch.cuda.current_device())
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    wa, w2, pd = d(st["w_avg"][None]), d(st["w2_avg"][None]), d(st["pre_D"][None])
    idx = torch.zeros(1, dtype=torch.int32, device="cuda")
    x32 = torch.tensor(inputs["slow"]).cuda()
    m.cuda()
    try:
        def reference_order(x, seed):
            torch.manual_seed(seed)
            z1 = torch.randn((1, 7583), device=g)
            z2 = torch.randn((30, 1), device=g)
            e1 = torch.randn(x.shape[0], 20, device=g)
            e2 = torch.randn(x.shape[0], 20, device=g)
            out = ops.multiswag(x, wa, w2, pd, idx, z1, z2.reshape(1, -1).contiguous(), torch.stack((e1, e2), 1)[None].contiguous())[0]
            return out, ops.swag_draw(wa, w2, pd, idx, z1, z2.reshape(1, -1).contiguous())[0], torch.randn(3, device=g)
        for x, seed in ((x32, 11), (x32, 12), (x32[:15].contiguous(), 13), (x32, 14)):
            want, w_want, next_want = reference_order(x, seed)
            torch.manual_seed(seed)
            got = m.forward_swag_fast(x, scale=0.5)
            nxt = torch.randn(3, device=g)                      # the generator stands where the reference's would
            assert got.is_cuda and torch.equal(got, want) and torch.equal(nxt, next_want), seed
            assert torch.equal(m.flatten().to(g), w_want), seed  # :838, re-drawn lazily from the call's own normals
        s2 = torch.cuda.Stream()
        s2.wait_stream(torch.cuda.current_stream())
        want, _, _ = reference_order(x32, 15)
        with torch.cuda.stream(s2):
            torch.manual_seed(15)
            got = m.forward_swag_fast(x32, scale=0.5)
        s2.synchronize()
        assert torch.equal(got, want)
        # a second model has buffers of its own: its call does not touch what the first one "has loaded"
        m2 = models[12]
        m2.cuda()
        torch.manual_seed(16)
        m.forward_swag_fast(x32, scale=0.5)
        _, w_want, _ = reference_order(x32, 16)
        torch.manual_seed(17)
        m2.forward_swag_fast(x32, scale=0.5)
        assert torch.equal(m.flatten().to(g), w_want)
        m2.cpu()
    finally:
        m.cpu()


def test_data_setup_kernel_drop_in():
    from bnn_chaos_model_amd import regression
    z = load_golden("case_features.npz")
    X = regression.data_setup_kernel(z["mass"][1], z["tseries"][1][None])
    assert X.shape == (1, 100, 41) and X.dtype == np.float64
    assert np.abs(X[0] - z["X64"][1]).max() <= 4.5e-16
    x = regression.pack_features(z["tseries"], z["mass"])
    assert x.is_cuda and x.dtype == torch.float32 and np.abs(x.cpu().numpy().astype(np.float64) - z["x32"]).max() <= 3e-6


def test_forward_with_random_sample_replays_reference_seed(models):
    """VarModel.forward with random_sample = True (`augment`, :404-408, :502-503): the same np.random.randint calls, the series the reference
    picked -- its length is whatever came up (85, 15, 27 here: the pretrained network's embedded forms of the generic engine) --, the same
    numbers under the same seeds; sample() switches the augmentation off for its loop and puts the flag back (:532-543)."""
    z = load_golden("case_augment.npz")
    m = models[0]
    m.load(torch.tensor(z["w"]))
    x = torch.tensor(z["x"])
    m.random_sample = True
    try:
        for i in range(int(z["runs"])):
            seed = int(z[f"run{i}_seed"])
            np.random.seed(seed)
            torch.manual_seed(seed)
            out = m(x, noisy_val=bool(int(z[f"run{i}_noisy"])))
            nbad, mx = close_report(out.numpy(), z[f"run{i}_out"])
            assert nbad == 0, (i, nbad, mx)
        np.random.seed(7200)
        torch.manual_seed(7200)
        s = m.sample(x, samples=2)
        assert m.random_sample is True
        want = z["sample_out"]
        assert np.abs(s - want).max() <= 2e-5 * np.abs(want).max()
    finally:
        m.random_sample = False


def test_torch_custom_ops(swag_states, inputs):
    """torch.ops.bnn_chaos.* (torch.library custom ops) give the same bits as the python wrappers; fake impls give shapes."""
    import bnn_chaos_model_amd.torch_ops  # noqa: F401  (registers the ops)
    from bnn_chaos_model_amd import ops
    st = swag_states[0]
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    wa, w2, pd = d(st["w_avg"][None]), d(st["w2_avg"][None]), d(st["pre_D"][None])
    x = d(inputs["slow"])
    idx = torch.zeros(3, dtype=torch.int32, device="cuda")
    a = torch.ops.bnn_chaos.multiswag(x, wa, w2, pd, idx, None, None, None, 1, 0.5, 42, 0, 0)
    b = ops.multiswag(x, wa, w2, pd, idx, philox_seed=42)
    assert torch.equal(a, b)
    W = torch.ops.bnn_chaos.swag_draw(wa, w2, pd, idx, None, None, 0.5, 42, 0)
    c = torch.ops.bnn_chaos.forward(x, W, None, None, None, 1, False, 42, 0, 0)
    assert torch.equal(a, c)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        fx = torch.empty((7, 100, 41), device="cuda")
        fo = torch.ops.bnn_chaos.multiswag(fx, torch.empty((1, 7583), device="cuda"), torch.empty((1, 7583), device="cuda"),
                                           torch.empty((1, 7583, 30), device="cuda"), torch.empty(6, dtype=torch.int32, device="cuda"),
                                           None, None, None, 2, 0.5, 0, 0, 0)
        assert fo.shape == (3, 7, 2)
    mom = torch.ops.bnn_chaos.multiswag_moments(x, wa, w2, pd, idx, 0.5, 42, 0, 0, 2)     # slabs of 2 draws
    assert torch.allclose(mom, ops.moments(b), rtol=1e-13, atol=0)
    t = torch.ops.bnn_chaos.multiswag_stats(x, wa, w2, pd, idx, 1, 0.5, 42, 0, 0)
    assert torch.equal(t, ops.stats_draw(b, philox_seed=42))


def test_torch_custom_ops_take_any_network(tmp_path):
    """A checkpoint built with other hparams (spock_reg_model.py:343-362: hidden 64, latent 16 -- the fixture case_arch_h64l16, written by
    the reference class) through ALL FIVE custom ops -- swag_draw, forward, multiswag, multiswag_moments, multiswag_stats take the
    network (`net`), its column mask and its clamp floor -- against ops.* on the same plan, bit for bit; and against the reference's own
    forward_swag_fast output through the module surface, which routes through these ops."""
    import bnn_chaos_model_amd.torch_ops  # noqa: F401
    from bnn_chaos_model_amd import checkpoint, ops
    from bnn_chaos_model_amd import spock_reg_model as srm
    z = load_golden("case_arch_h64l16.npz")
    hp = json.loads(str(z["hparams_json"]))
    for k, v in list(hp.items()):
        if isinstance(v, str) and v in ("True", "False"):
            hp[k] = v == "True"
    path = str(tmp_path / "net64_output.pkl")
    checkpoint.write_swag_file(path, hp, json.loads(str(z["swa_params_json"])), torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"]))
    m = srm.load_swag(path).cpu().eval()
    mask, lowest, net = m._op_args()
    assert net[:5] == [41, 64, 16, 1, 1]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plan = m._plan()
    assert not plan.v50net
    if not (plan.spec_attached(False) or plan.spec_attached(True)):   # (another test of this session may have compiled this network's form)
        plan.__dict__.pop("_warned_generic", None)
        with pytest.warns(srm.GenericEngineWarning, match="specialize"):   # said once per plan, with the expected fraction and the way out
            m._plan()
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            m._plan()
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    wa, w2, pd = d(z["w_avg"][None]), d(z["w2_avg"][None]), d(z["pre_D"][None])
    x = d(z["x"])
    idx = torch.zeros(4, dtype=torch.int32, device="cuda")
    T = torch.ops.bnn_chaos
    W = T.swag_draw(wa, w2, pd, idx, None, None, 0.5, 42, 0, mask, lowest, net)
    assert W.shape == (4, plan.d) and torch.equal(W, ops.swag_draw(wa, w2, pd, idx, philox_seed=42, plan=plan))
    a = T.multiswag(x, wa, w2, pd, idx, None, None, None, 1, 0.5, 42, 0, 0, mask, lowest, net)
    b = ops.multiswag(x, wa, w2, pd, idx, philox_seed=42, plan=plan)
    assert torch.equal(a, b) and torch.equal(a, T.forward(x, W, None, None, None, 1, False, 42, 0, 0, mask, lowest, net))
    mom = T.multiswag_moments(x, wa, w2, pd, idx, 0.5, 42, 0, 0, 2, mask, lowest, net)
    assert torch.equal(mom, ops.multiswag_moments(x, wa, w2, pd, idx, philox_seed=42, draws_per_launch=2, plan=plan))
    assert torch.allclose(mom, ops.moments(b), rtol=1e-13, atol=0)
    t = T.multiswag_stats(x, wa, w2, pd, idx, 1, 0.5, 42, 0, 0, mask, lowest, net)
    assert torch.equal(t, ops.stats_draw(b, philox_seed=42)) and torch.equal(t, ops.multiswag_stats(x, wa, w2, pd, idx, philox_seed=42, plan=plan))
    # without the network arguments these ops would take the pretrained network's plan: refused on the parameter count, never silently wrong
    with pytest.raises((ValueError, RuntimeError)):
        T.multiswag_moments(x, wa, w2, pd, idx, 0.5, 42, 0, 0, 2)
    # assume_finite is part of the four x-reading ops' schema (default False: x is scanned); a damaged system is NaN through them
    xb = x.clone(); xb[1, 5, 3] = float("inf")       # a masked column: NaN after the reference's x - mask
    for got in (T.multiswag(xb, wa, w2, pd, idx, None, None, None, 1, 0.5, 42, 0, 0, mask, lowest, net)[:, :, 0],
                T.forward(xb, W, None, None, None, 1, False, 42, 0, 0, mask, lowest, net)[:, :, 0],
                T.multiswag_moments(xb, wa, w2, pd, idx, 0.5, 42, 0, 0, 2, mask, lowest, net)[None, :, 0].float(),
                T.multiswag_stats(xb, wa, w2, pd, idx, 1, 0.5, 42, 0, 0, mask, lowest, net)):
        assert torch.isnan(got[:, 1]).all() and torch.isfinite(got[:, 0]).all() and torch.isfinite(got[:, 2:]).all()
    assert torch.isfinite(T.multiswag(xb, wa, w2, pd, idx, None, None, None, 1, 0.5, 42, 0, 0, mask, lowest, net, True)[:, 0]).all()
    # the module surface on this checkpoint = these ops: the reference's own numbers
    tp = [z[f"swagfast_tape_{i:03d}"] for i in range(int(z["swagfast_tape_n"]))]
    torch.manual_seed(hp["seed"] + 2)
    out = m.forward_swag_fast(torch.tensor(z["x"]), scale=0.5)
    nbad, mx = close_report(out.numpy(), z["swagfast_out"])
    assert nbad == 0, (nbad, mx)


def test_statistics_epilogue_replays_reference():
    """fast_truncnorm -> prior resampling -> min over trios -> percentiles, with numpy's generator consumed as the
    reference consumes it: bit-identical to the captured reference fragments (figures/multiswag_5_planet.py:306-428)."""
    from bnn_chaos_model_amd import stats
    z = load_golden("case_stats.npz")
    np.random.seed(6000)
    tn = stats.fast_truncnorm(z["loc"], z["scale"], left=4, d=int(z["d"]), nsamp=int(z["nsamp"]), seed=0)
    assert tn.is_cuda and tn.shape == z["truncnorm"].shape and np.array_equal(tn.cpu().numpy(), z["truncnorm"])
    assert stats.prior_normalization() == float(z["normalization"])
    np.random.seed(6001)
    rs = stats.resample_prior(tn)
    assert np.array_equal(rs.cpu().numpy(), z["resampled"])
    outs = stats.min_over_trios(rs)
    assert np.array_equal(outs.cpu().numpy(), z["outs"])
    q = [50.0, 50 + 68 / 2, 50 - 68 / 2, 50 + 95 / 2, 50 - 95 / 2]
    pc = stats.percentiles(outs, q).cpu().numpy()
    want = np.stack([np.percentile(z["outs"][i].astype(np.float64), q) for i in range(z["outs"].shape[0])])
    assert np.abs(pc - want).max() <= 1e-6
    # (mu, std) tensor form and in-kernel noise: same distribution (every sample > left unless nothing passes)
    musd = torch.stack([torch.tensor(z["loc"]), torch.tensor(z["scale"])], -1)
    ph = stats.fast_truncnorm(musd, left=4, nsamp=40, seed=3, rng="philox").cpu().numpy()
    assert ph.shape == z["loc"].shape and (ph[1:] > 4).mean() > 0.999 and (ph[0, :5] < 4).all()
    rp = stats.resample_prior(torch.tensor(ph), rng="philox", seed=4).cpu().numpy()
    moved = ph >= 9
    assert (rp[moved] >= 9).all() and (rp[moved] <= 100).all() and np.array_equal(rp[~moved], ph[~moved])
    two = stats.fast_truncnorm(musd, left=4, right=12, nsamp=40, seed=3, rng="philox").cpu().numpy()   # two-sided, in-kernel noise
    inside = (two > 4) & (two < 12)
    assert inside[1:].mean() > 0.99 and np.array_equal(two[ph < 12], ph[ph < 12])   # same candidates, stricter acceptance


def test_sample_tseries_equals_the_reference_loop(ckpt_dir):
    """FeatureRegressor.sample downstream of REBOUND (regression.py:137-162): one launch per trio == the literal loop."""
    from bnn_chaos_model_amd import regression
    from bnn_chaos_model_amd.regression import FeatureRegressor
    z = load_golden("case_features.npz")
    fr = FeatureRegressor(cuda=False, filebase=str(ckpt_dir / "*v50*output.pkl"), sort=True)
    ts = np.repeat(z["tseries"][:2], 10, axis=1)  # [2 trios, 1000, 26]: every 10th row is the fixture's series
    np.random.seed(11); torch.manual_seed(11)
    mu, std = fr.sample_tseries(ts, z["mass"][:2], samples=5)
    assert mu.shape == std.shape == (2, 5)
    np.random.seed(11); torch.manual_seed(11)
    want = []
    for i in range(2):  # regression.py:137-150, literally
        X = regression.data_setup_kernel(z["mass"][i], ts[None, i, ::10])
        X = fr.ssX.transform(X.reshape(-1, X.shape[-1])).reshape(X.shape)
        X = torch.tensor(X).float()
        want.append(torch.cat([fr.sample_full_swag(X)[None] for _ in range(5)], dim=0).detach().numpy())
    want = np.array(want)[..., 0, :]
    assert np.abs(mu - want[..., 0]).max() <= 1e-5 and np.abs(std - want[..., 1]).max() <= 1e-5


def test_five_planet_pipeline_end_to_end(ckpt_dir):
    """features -> MultiSWAG MC loop -> truncnorm -> prior resampling -> min over trios -> percentile bands, all on the GPU."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("five_planet_pipeline", os.path.join(ROOT, "examples", "five_planet_pipeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    np.random.seed(0)
    r = mod.run(str(ckpt_dir / "*v50*output.pkl"), sims=12, samples=40, rng="philox", seed=5)
    assert r["time"].shape == (40, 12, 3, 2) and torch.isfinite(r["time"]).all()
    assert r["outs"].shape == (12, 40) and r["bands"].shape == (12, 5)
    b = r["bands"].cpu().numpy()
    assert (b[:, 4] <= b[:, 2]).all() and (b[:, 2] <= b[:, 0]).all() and (b[:, 0] <= b[:, 1]).all() and (b[:, 1] <= b[:, 3]).all()
    s = r["samps_time"].cpu().numpy()
    assert ((s > 4) | (s < 4.0001)).all() and s.max() <= 100
    np.random.seed(0)
    r2 = mod.run(str(ckpt_dir / "*v50*output.pkl"), sims=12, samples=40, rng="philox", seed=5)
    assert torch.equal(r["bands"], r2["bands"])  # counter-based noise: reproducible end to end
    # the streaming form (statistics fused behind the forward kernel + quantile sketch): same distribution, different Philox
    # streams; with 400 samples the medians of the two forms agree to a few standard errors
    a = mod.run(str(ckpt_dir / "*v50*output.pkl"), sims=12, samples=400, rng="philox", seed=5)
    st = mod.run(str(ckpt_dir / "*v50*output.pkl"), sims=12, samples=400, rng="philox", seed=5, streaming=True)
    assert st["bands"].shape == (12, 5) and torch.isfinite(st["bands"]).all()
    sb = st["bands"].cpu().numpy()
    assert (sb[:, 4] <= sb[:, 2]).all() and (sb[:, 2] <= sb[:, 0]).all() and (sb[:, 0] <= sb[:, 1]).all() and (sb[:, 1] <= sb[:, 3]).all()
    spread = (a["bands"][:, 1] - a["bands"][:, 2]).cpu().numpy() / 2 + 0.05     # ~1 sigma of the predictive distribution
    assert (np.abs(sb[:, 0] - a["bands"][:, 0].cpu().numpy()) < 5 * 1.25 * spread / np.sqrt(400) + 0.02).all()
    assert (np.abs(st["average"].cpu().numpy() - a["average"].cpu().numpy()) < 5 * spread / np.sqrt(400) + 0.3).all()


def test_integration_bindings_run_the_callers_lines(ckpt_dir, inputs, monkeypatch):
    """INTEGRATION.md section 1, executed: with the shim directory ahead on sys.path, the literal caller lines of
    figures/multiswag_5_planet.py (:61, :257, :280-298) and figures/main_figures.py (:39-42, :127-139, :154-156) run
    unchanged and reproduce the runs captured from the reference (case_chunk_loop.npz, case_multiswag_grid.npz)."""
    import glob
    import os
    import sys
    import bnn_chaos_model_amd
    monkeypatch.syspath_prepend(os.path.join(os.path.dirname(bnn_chaos_model_amd.__file__), "shims"))
    for name in ("spock", "spock_reg_model"):
        monkeypatch.delitem(sys.modules, name, raising=False)
    import spock_reg_model                                           # main_figures.py: `import spock_reg_model`
    from spock import FeatureRegressor, FeatureRegressorXGB          # multiswag_5_planet.py:29  # noqa: F401

    # ---- figures/multiswag_5_planet.py
    version = 50
    pretrained = str(ckpt_dir) + "/"
    names = sorted(glob.glob(pretrained + "*" + f"v{version:d}" + "*output.pkl"))
    model = FeatureRegressor(                                        # :61-66 (the reference's glob order is unsorted; the
        cuda=False,                                                  #  fixture was captured with CPU generators and sorted seeds)
        filebase=pretrained + "*" + f"v{version:d}" + "*output.pkl",
        sort=True)
    assert [m.hparams["seed"] for m in model.swag_ensemble] == [spock_reg_model.load_swag(n).hparams["seed"] for n in names]
    rng = np.random.default_rng(0)
    X = model.ssX.inverse_transform(rng.standard_normal((4, 3, 100, 41)))   # un-standardised features [sim, trio, time, feature]
    allmeg = X[..., model.swag_ensemble[0].megno_location].ravel()   # :257
    assert allmeg.shape == (4 * 3 * 100,)
    Xp = (model.ssX                                                  # :280-283
          .transform(X.reshape(-1, X.shape[-1]))
          .reshape(X.shape)
          )
    Xpp = torch.tensor(Xp).float()                                   # :287
    Xflat = Xpp.reshape(-1, X.shape[-2], X.shape[-1])                # :289
    if model.cuda:                                                   # :291-292
        Xflat = Xflat.cuda()
    assert Xflat.shape == (12, 100, 41) and Xflat.dtype == torch.float32
    z = load_golden("case_chunk_loop.npz")
    Xflat = torch.tensor(inputs["slow"][:30])                        # the captured run's inputs
    samples = int(z["samples"])
    np.random.seed(5000)
    torch.manual_seed(5000)
    time = torch.cat([                                               # :295-298
        torch.cat([model.sample_full_swag(Xpart).detach().cpu() for Xpart in torch.chunk(Xflat, chunks=10)])[None]
        for _ in range(samples)
    ], dim=0).reshape(samples, 10, 3, 2).numpy()
    nbad, mx = close_report(time.reshape(samples, 30, 2), z["out"])
    assert nbad == 0, (nbad, mx)

    # ---- figures/main_figures.py
    checkpoint_filename = "v50"
    swag_ensemble = [                                                # :39-42 (sorted, CPU generators: as captured)
        spock_reg_model.load_swag(fname)
        for fname in sorted(glob.glob(pretrained + "*" + checkpoint_filename + "*output.pkl"))
    ]

    def sample_full_swag(X_sample, gpu):                             # :127-139
        swag_i = np.random.randint(0, len(swag_ensemble))
        swag_model = swag_ensemble[swag_i]
        swag_model.eval()
        if gpu:
            swag_model.w_avg = swag_model.w_avg.cuda()
            swag_model.w2_avg = swag_model.w2_avg.cuda()
            swag_model.pre_D = swag_model.pre_D.cuda()
            swag_model.cuda()
        out = swag_model.forward_swag(X_sample, scale=0.5)
        return out

    g = load_golden("case_multiswag_grid.npz")
    X_sample = torch.tensor(inputs["slow"])
    np.random.seed(4000)
    torch.manual_seed(4000)
    raw = np.array([sample_full_swag(X_sample, False).cpu().detach().numpy() for _ in range(3)])   # :154-156 (2000 there)
    nbad, mx = close_report(raw, g["out"])
    assert nbad == 0, (nbad, mx)
    # the same lines with the models and the batch on the GPU, as the script has them: noise then comes from the GPU generator
    X_sample = X_sample.cuda()
    raw = np.array([sample_full_swag(X_sample, True).cpu().detach().numpy() for _ in range(3)])
    assert raw.shape == (3, 32, 2) and np.isfinite(raw).all() and raw[..., 0].min() >= 4 and raw[..., 1].max() <= 6
    _preds = np.concatenate([raw], axis=1)
    assert np.median(_preds[..., 0], 0).shape == (32,)              # :277-278


def test_fast_truncnorm_two_sided_and_right_sided():
    """stats.fast_truncnorm with a finite right bound / an open left bound replays the reference (numpy generator consumed in
    chunks of d elements, as the reference does) bit for bit."""
    from bnn_chaos_model_amd import stats
    z = load_golden("case_truncnorm2.npz")
    for name in ("two_sided", "right_only"):
        np.random.seed(6100)
        got = stats.fast_truncnorm(z["loc"], z["scale"], left=float(z[f"{name}_left"]), right=float(z[f"{name}_right"]),
                                   d=int(z["d"]), nsamp=int(z["nsamp"])).cpu().numpy()
        assert np.array_equal(got, z[f"{name}_out"]), name
