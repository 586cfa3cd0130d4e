"""Edge cases of the HIP path against the oracle: other T / K / masks / clamp floors, empty inputs, bad seed indices,
chunk partitions with short and missing chunks.  Needs an MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, close_report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from bnn_chaos_model_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as _orc
    return _orc


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def synth(B, T, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, 1, 41)) + 0.1 * rng.standard_normal((B, T, 41))).astype(np.float32)
    x[:, :, 0] = np.linspace(-1.71, 1.74, T, dtype=np.float32)[None]
    return x


def sched(ops, orc, plan, noisy=False):
    return orc.make_schedule([plan.layer_order(l, noisy) for l in range(6)], pool_parts=4)


def state(swag_states, K=30):
    st = swag_states[0]
    pd = st["pre_D"][:, :K]
    return st["w_avg"][None], st["w2_avg"][None], np.ascontiguousarray(pd)[None]


@pytest.mark.parametrize("T", (8, 12, 40, 100, 104, 400))
def test_other_sequence_lengths(T, ops, orc, swag_states):
    B, J = 21, 2
    x = synth(B, T, T)
    rng = np.random.default_rng(T)
    wa, w2, pd = state(swag_states)
    z1 = rng.standard_normal((J, 7583), dtype=np.float32); z2 = rng.standard_normal((J, 30), dtype=np.float32)
    eps = rng.standard_normal((J, B, 2, 20), dtype=np.float32)
    idx = np.zeros(J, np.int32)
    out, pre, summ = ops.multiswag(dev(x), dev(wa), dev(w2), dev(pd), torch.as_tensor(idx), dev(z1), dev(z2), dev(eps), debug=True)
    plan = ops.get_plan()
    o = orc.multiswag(x, wa, w2, pd, idx, z1, z2, eps, arch=orc.make_arch(T=T), sched=sched(ops, orc, plan))
    assert np.abs(out.cpu().numpy() - o).max() <= 2e-6
    # T % 4 != 0 (and T < 8): the generic engine, same answer as the oracle on ITS schedule (tests/test_hip_arch.py has the reference fixtures)
    x10 = synth(B, 10, 3)
    o10 = ops.multiswag(dev(x10), dev(wa), dev(w2), dev(pd), torch.as_tensor(idx), dev(z1), dev(z2), dev(eps))
    w10 = orc.multiswag(x10, wa, w2, pd, idx, z1, z2, eps, arch=orc.make_arch(T=10), sched=orc.make_schedule(None, pool_parts=4))
    assert np.abs(o10.cpu().numpy() - w10).max() <= 2e-6
    with pytest.raises(Exception):
        ops.multiswag(dev(synth(2, 1, 0)), dev(wa), dev(w2), dev(pd), torch.as_tensor(idx))  # T = 1: torch.std is NaN


@pytest.mark.parametrize("K", (2, 7, 20, 32))
def test_other_swag_ranks(K, ops, orc, swag_states):
    rng = np.random.default_rng(K)
    st = swag_states[12]
    pd = (np.tile(st["pre_D"], (1, 2))[:, :K]).copy()
    wa, w2, pd = st["w_avg"][None], st["w2_avg"][None], pd[None]
    J = 3
    z1 = rng.standard_normal((J, 7583), dtype=np.float32); z2 = rng.standard_normal((J, K), dtype=np.float32)
    W = ops.swag_draw(dev(wa), dev(w2), dev(pd), torch.zeros(J, dtype=torch.int32), dev(z1), dev(z2), scale=0.7).cpu().numpy()
    for j in range(J):
        assert np.array_equal(W[j], orc.swag_draw(wa[0], w2[0], pd[0], z1[j], z2[j], scale=0.7))
    # in-kernel draw (single launch) agrees with the draw kernel for this K too
    x = synth(5, 100, K)
    eps = rng.standard_normal((J, 5, 2, 20), dtype=np.float32)
    a = ops.multiswag(dev(x), dev(wa), dev(w2), dev(pd), torch.zeros(J, dtype=torch.int32), dev(z1), dev(z2), dev(eps), scale=0.7, single_launch=True)
    b = ops.forward(dev(x), dev(W), eps=dev(eps))
    assert torch.equal(a, b)
    with pytest.raises(Exception):
        ops.swag_draw(dev(wa), dev(w2), dev(np.zeros((1, 7583, 257), np.float32)), torch.zeros(1, dtype=torch.int32))   # K <= 256


@pytest.mark.parametrize("flags", [dict(fix_megno2=False, include_mmr=True, include_nan=True, include_eplusminus=True),
                                   dict(fix_megno2=True, include_mmr=True, include_nan=False, include_eplusminus=True),
                                   dict(fix_megno2=True, include_mmr=False, include_nan=False, include_eplusminus=False)])
@pytest.mark.parametrize("lowest", (0.5, 0.1))
def test_other_masks_and_clamp_floor(flags, lowest, ops, orc, swag_states):
    """Any column mask runs on the 41-column engine; lower_std moves the soft_clamp floor (spock_reg_model.py:363-365)."""
    mask = ops.zero_mask_from_flags(**flags)
    plan = ops.get_plan(mask, lowest)
    B = 19
    x = synth(B, 100, 3)
    rng = np.random.default_rng(9)
    w = (swag_states[0]["w_avg"] + 0.05 * rng.standard_normal(7583)).astype(np.float32)
    eps = rng.standard_normal((1, B, 2, 20), dtype=np.float32)
    out, pre, summ = ops.forward(dev(x), dev(w[None]), eps=dev(eps), plan=plan, debug=True)
    arch = orc.make_arch(T=100, zero_mask=mask, lowest=lowest)
    o, ex = orc.forward(x, w, eps[0, :, 0], eps[0, :, 1], arch=arch, sched=sched(ops, orc, plan), extras=True)
    assert np.array_equal(summ.cpu().numpy()[0], ex["summary"])
    assert np.array_equal(pre.cpu().numpy()[0], ex["pre_clamp"])
    assert np.abs(out.cpu().numpy()[0] - o).max() <= 2e-6
    assert out[..., 1].min().item() >= lowest


def test_noisy_forward_generic_mask(ops, orc, swag_states):
    mask = ops.zero_mask_from_flags(fix_megno2=True, include_mmr=True, include_nan=False, include_eplusminus=True)
    plan = ops.get_plan(mask, 0.5)
    B = 9
    x = synth(B, 100, 5)
    rng = np.random.default_rng(5)
    w = swag_states[12]["w_avg"]
    eps = rng.standard_normal((1, B, 2, 20), dtype=np.float32)
    e_in = rng.standard_normal((1, B, 100, 41), dtype=np.float32)
    e_sum = rng.standard_normal((1, B, 40), dtype=np.float32)
    out = ops.forward(dev(x), dev(w[None]), eps=dev(eps), eps_in=dev(e_in), eps_sum=dev(e_sum), plan=plan).cpu().numpy()[0]
    o = orc.forward(x, w, eps[0, :, 0], eps[0, :, 1], eps_in=e_in[0], eps_sum=e_sum[0], arch=orc.make_arch(T=100, zero_mask=mask),
                    sched=sched(ops, orc, plan, noisy=True))
    nbad, mx = close_report(out, o, rtol=2e-6, atol=2e-6)  # expf differs by an ulp between libm and the device
    assert nbad == 0, mx


def test_empty_and_degenerate_shapes(ops, swag_states):
    wa, w2, pd = (dev(a) for a in state(swag_states))
    x0 = torch.zeros((0, 100, 41), device="cuda")
    out = ops.multiswag(x0, wa, w2, pd, torch.zeros(3, dtype=torch.int32))
    assert out.shape == (3, 0, 2)
    out = ops.multiswag(dev(synth(4, 100, 1)), wa, w2, pd, torch.zeros(0, dtype=torch.int32))
    assert out.shape == (0, 4, 2)
    W = ops.swag_draw(wa, w2, pd, torch.zeros(0, dtype=torch.int32))
    assert W.shape == (0, 7583)


def test_bad_seed_index_poisons_only_its_draw(ops, swag_states):
    wa, w2, pd = (dev(a) for a in state(swag_states))
    x = dev(synth(6, 100, 2))
    idx = torch.tensor([0, 5, 0, -1], dtype=torch.int32)
    for single in (True, False):
        out = ops.multiswag(x, wa, w2, pd, idx, philox_seed=3, single_launch=single).cpu().numpy()
        assert np.isfinite(out[0]).all() and np.isfinite(out[2]).all()
        assert np.isnan(out[1]).all() and np.isnan(out[3]).all()


@pytest.mark.parametrize("B,chunks", [(30, 10), (25, 10), (7, 10), (1, 10), (1000, 3), (64, 64)])
def test_chunk_partitions_match_torch_chunk(B, chunks, ops, orc, swag_states):
    """torch.chunk gives ceil(B/chunks)-sized chunks, a short last one, and fewer chunks than asked when B is small."""
    parts = torch.chunk(torch.arange(B), chunks)
    nch = len(parts)
    samples = 2
    J = samples * nch
    rng = np.random.default_rng(B)
    x = synth(B, 100, B)
    wa, w2, pd = state(swag_states)
    z1 = rng.standard_normal((J, 7583), dtype=np.float32); z2 = rng.standard_normal((J, 30), dtype=np.float32)
    eps = rng.standard_normal((samples, B, 2, 20), dtype=np.float32)
    idx = np.zeros(J, np.int32)
    out = ops.multiswag(dev(x), dev(wa), dev(w2), dev(pd), torch.as_tensor(idx), dev(z1), dev(z2), dev(eps), nchunks=nch).cpu().numpy()
    want = np.zeros((samples, B, 2), np.float32)
    plan = ops.get_plan()
    for e in range(J):  # the reference loop, chunk by chunk, through the oracle
        s, c = divmod(e, nch)
        rows = parts[c].numpy()
        w = orc.swag_draw(wa[0], w2[0], pd[0], z1[e], z2[e])
        want[s, rows] = orc.forward(x[rows], w, eps[s, rows, 0], eps[s, rows, 1], sched=sched(ops, orc, plan))
    assert np.abs(out - want).max() <= 2e-6


def test_noisy_forward_in_kernel_noise_equals_explicit(ops, swag_states):
    """forward(noisy_val=True) with every normal generated in-kernel == the same call fed the generated normals explicitly."""
    B, R, seed = 37, 3, 77
    x = dev(synth(B, 100, 8))
    W = dev(np.stack([swag_states[0]["w_avg"], swag_states[12]["w_avg"], swag_states[0]["w_avg"]]))
    a = ops.forward(x, W, philox_seed=seed, draw_id0=4, system_id0=1000, noisy=True)
    eps = ops.philox_normal(2, seed, 4, R, B=B, system_id0=1000)
    e_in = ops.philox_normal(3, seed, 4, R, width=100, B=B, system_id0=1000)
    e_sum = ops.philox_normal(4, seed, 4, R, B=B, system_id0=1000)
    b = ops.forward(x, W, eps=eps, eps_in=e_in, eps_sum=e_sum)
    assert torch.equal(a, b)
    # sharding invariance and plausibility of the input-noise normals
    c = ops.forward(x[20:].contiguous(), W, philox_seed=seed, draw_id0=4, system_id0=1020, noisy=True)
    assert torch.equal(a[:, 20:], c)
    n = e_in.flatten().double().cpu().numpy()
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01
    quiet = ops.forward(x, W, philox_seed=seed, draw_id0=4, system_id0=1000)
    assert not torch.equal(a, quiet)


def test_input_noise_stream_statistics(ops):
    """The input-noise stream (TAG_IN) runs Philox4x32 with 7 rounds and cuts each block into six 21-bit uniforms: 3.3e6 of its
    normals pass a Kolmogorov-Smirnov test against N(0,1), have the right moments, and show no correlation between neighbouring
    columns (same block), neighbouring timesteps / systems / draws (neighbouring counters)."""
    from scipy import stats
    R, B, T = 4, 200, 100
    e = ops.philox_normal(3, 2024, 17, R, width=T, B=B, system_id0=5_000_000).double().cpu().numpy()   # [R,B,T,41]
    n = e.ravel()
    assert n.size == R * B * T * 41
    ks = stats.kstest(n[::7], "norm")                        # a 4.7e5-sample thinning keeps the test's own resolution sensible
    assert ks.pvalue > 1e-3, ks
    assert abs(n.mean()) < 3e-3 and abs(n.std() - 1) < 3e-3
    assert abs(stats.skew(n)) < 6e-3 and abs(stats.kurtosis(n)) < 1.2e-2   # 4 sigma of sqrt(6/N), sqrt(24/N)
    assert np.abs(n).max() < 5.5 and np.abs(n).max() > 4.2   # 21-bit radius levels: reaches 5.4 sigma, no further

    def corr(a, b):
        return abs(np.corrcoef(a.ravel(), b.ravel())[0, 1])
    bound = 4.0 / np.sqrt(n.size / 2)
    assert corr(e[..., 0:40:2], e[..., 1:41:2]) < bound      # cos / sin partners and block neighbours
    assert corr(e[..., :-1], e[..., 1:]) < bound
    assert corr(e[:, :, :-1], e[:, :, 1:]) < bound           # consecutive timesteps
    assert corr(e[:, :-1], e[:, 1:]) < bound                 # consecutive systems
    assert corr(e[:-1], e[1:]) < bound                       # consecutive output rows
    assert corr(e ** 2, np.roll(e, 1, axis=-1) ** 2) < bound  # no dependence in the magnitudes either


def test_sample_with_philox_rng(swag_states, tmp_path):
    """VarModel.sample with rng='philox': same estimator, in-kernel noise; agrees statistically with the torch-rng path."""
    import json
    from bnn_chaos_model_amd import checkpoint, spock_reg_model as srm
    from conftest import load_golden
    z = load_golden("swag_v50_0.npz")
    p = tmp_path / "m_v50_0_output.pkl"
    checkpoint.write_swag_file(str(p), json.loads(str(z["hparams_json"])), json.loads(str(z["swa_params_json"])),
                               torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"]))
    m = srm.load_swag(str(p))
    m.load(m.w_avg)
    x = torch.tensor(synth(16, 100, 4))
    N = 3200
    torch.manual_seed(0); np.random.seed(0)
    a = m.sample(x, samples=N)
    m.rng, m.philox_seed = "philox", 5
    np.random.seed(1)
    b = m.sample(x, samples=N)
    assert a.shape == b.shape == (16,)
    # each is a mean of N draws of mu + n * std with std <= 6: the SE of the difference is <= sqrt(2) * 6 / sqrt(N) = 0.15,
    # so 0.6 is a 4-sigma bound for the worst possible system
    assert np.abs(a - b).max() < 0.6, np.abs(a - b).max()
    # the philox form is ONE launch for all samples and equals the sample-by-sample loop of noisy forwards bit for bit
    m._philox_calls = 100
    np.random.seed(2)
    one = m.sample(x, samples=7)
    m._philox_calls = 100
    np.random.seed(2)
    acc = []
    for _ in range(7):
        o = m(x).detach().cpu().numpy()
        acc.append(o[:, 0] + np.random.randn(16) * o[:, 1])
    assert np.array_equal(one, np.average(acc, axis=0))


def test_feature_pack_matches_reference(ops):
    """data_setup_kernel + ssX.transform + .float() on the GPU vs the captured reference output."""
    from conftest import load_golden
    from oracle import features
    z = load_golden("case_features.npz")
    x32, x64 = ops.feature_pack(z["tseries"], z["mass"], mean=z["mean"], scale=z["scale"], want_x64=True)
    x64, x32 = x64.cpu().numpy(), x32.cpu().numpy()
    # float64: exact except cos/sin (device libm vs glibc, <= 2 ulp)
    assert np.abs(x64 - z["X64"]).max() <= 4.5e-16
    ang = [11, 12, 13, 14, 15, 16, 20, 21, 22, 23, 24, 25, 29, 30, 31, 32, 33, 34]
    rest = [c for c in range(41) if c not in ang]
    assert np.array_equal(x64[..., rest], z["X64"][..., rest])
    # float32 network input: identical up to one rounding of those ulps
    assert np.abs(x32.astype(np.float64) - z["x32"]).max() <= 3e-7 * max(1.0, np.abs(z["x32"]).max())
    assert np.array_equal(x32[..., rest], z["x32"][..., rest])
    # standardise-only entry (already packed X), and the numpy oracle
    y32 = ops.feature_pack(X=z["X64"], mean=z["mean"], scale=z["scale"]).cpu().numpy()
    assert np.array_equal(y32, z["x32"]) and np.array_equal(y32, features.standardize(z["X64"], z["mean"], z["scale"]))
    with pytest.raises(NotImplementedError):
        ops.feature_pack(np.zeros((1, 100, 25)), np.zeros((1, 3)), mean=z["mean"], scale=z["scale"])


@pytest.mark.parametrize("R", (1, 2, 7, 100, 2000, 3000, 4097))
def test_quantiles_match_numpy(R, ops):
    """np.median / np.percentile over the draw axis (figures/main_figures.py:277-278, multiswag_5_planet.py:484-489)."""
    rng = np.random.default_rng(R)
    B = 33
    s = rng.standard_normal((R, B, 2)).astype(np.float32)
    s[:, 0, 0] = 4.0          # ties
    if R > 3:
        s[1, 1, 1] = s[2, 1, 1]
    q = [50.0, 50 + 68 / 2, 50 - 68 / 2, 50 + 95 / 2, 50 - 95 / 2, 0.0, 100.0]
    got = ops.quantiles(dev(s), q).cpu().numpy()
    want = np.percentile(s.astype(np.float64), q, axis=0)            # [nq, B, 2]
    assert np.abs(got - np.moveaxis(want, 0, -1)).max() <= 2.4e-7 * 4
    assert np.abs(got[..., 0] - np.median(s, axis=0)).max() <= 2.4e-7 * 4
    with pytest.raises(Exception):
        ops.quantiles(torch.zeros((16385, 1, 2), device="cuda"))


@pytest.mark.parametrize("kind", ("zeros", "huge", "tiny", "constant_rows", "alternating"))
def test_degenerate_inputs_match_oracle(kind, ops, orc, swag_states):
    """Zero variance over time, saturated clamps, denormal-scale inputs: still bit-identical to the pinned oracle."""
    B = 18
    rng = np.random.default_rng(3)
    x = synth(B, 100, 12)
    if kind == "zeros":
        x[:] = 0
    elif kind == "huge":
        x *= 3e4
    elif kind == "tiny":
        x *= 1e-30
    elif kind == "constant_rows":
        x[:] = x[:, :1]
    elif kind == "alternating":
        x[:, ::2] += 5.0
    wa, w2, pd = state(swag_states)
    J = 2
    z1 = rng.standard_normal((J, 7583), dtype=np.float32); z2 = rng.standard_normal((J, 30), dtype=np.float32)
    eps = rng.standard_normal((J, B, 2, 20), dtype=np.float32)
    idx = np.zeros(J, np.int32)
    out, pre, summ = ops.multiswag(dev(x), dev(wa), dev(w2), dev(pd), torch.as_tensor(idx), dev(z1), dev(z2), dev(eps), debug=True)
    plan = ops.get_plan()
    sc = sched(ops, orc, plan)
    for j in range(J):
        w = orc.swag_draw(wa[0], w2[0], pd[0], z1[j], z2[j])
        o, ex = orc.forward(x, w, eps[j, :, 0], eps[j, :, 1], sched=sc, extras=True)
        assert np.array_equal(summ.cpu().numpy()[j], ex["summary"])
        assert np.array_equal(pre.cpu().numpy()[j], ex["pre_clamp"])
        assert np.abs(out.cpu().numpy()[j] - o).max() <= 2e-6
    assert torch.isfinite(out).all()


def test_non_finite_input_gets_the_reference_answer_and_stays_inside_its_own_system(ops, swag_states):
    """The reference returns NaN for a system that holds NaN anywhere or +-inf in a masked column (`x = x - mask`,
    spock_reg_model.py:452-478; NaN-propagating nn.ReLU): so does every op by default (ops.nonfinite_scan; the fixture test is
    tests/test_hip_nonfinite.py).  And a non-finite value never leaks into the other systems of its wave / workgroup."""
    B = 40
    x = synth(B, 100, 21)
    wa, w2, pd = state(swag_states)
    idx = torch.zeros(3, dtype=torch.int32)
    clean = ops.multiswag(dev(x), dev(wa), dev(w2), dev(pd), idx, philox_seed=9)
    bad = x.copy()
    bad[5, 17, 9] = np.nan          # live column
    bad[22, 3, 12] = np.inf         # live column
    bad[30, :, 3] = np.nan          # masked column (v50): the kernels never read it; the reference's x - x turns it into NaN
    got = ops.multiswag(dev(bad), dev(wa), dev(w2), dev(pd), idx, philox_seed=9)
    assert torch.isnan(got[:, [5, 22, 30]]).all()
    keep = [b for b in range(B) if b not in (5, 22, 30)]
    assert torch.equal(got[:, keep], clean[:, keep])
    blind = ops.multiswag(dev(bad), dev(wa), dev(w2), dev(pd), idx, philox_seed=9, assume_finite=True)   # the kernels alone
    assert torch.equal(blind[:, keep + [30]], clean[:, keep + [30]])


@pytest.mark.parametrize("record", ("default route", "caller-owned"))
def test_hip_graph_capture_and_replay(record, ops, swag_states):
    """The ops only enqueue work on the current stream -- no synchronisation, and no allocation on REPLAY: the draw workspace and, on the
    default route, the scan record are allocated while capturing, from the capturing graph's memory pool (ops._record_buffer says why);
    with `nonfinite=` + nonfinite_scan(out=...) the record is the caller's tensor and no record is allocated at all.  The two-launch
    multiswag call is captured once and replayed on the same buffers with clean -> damaged -> clean -> damaged inputs (NaN in a masked
    column: the direct NaN path; +inf in a live column: the exact re-evaluation): every replay must equal the eager call on the same x,
    i.e. the record's header is rebuilt by every replay -- 0 -> 2 -> 0 -> 2 listed systems -- and never carried over."""
    wa, w2, pd = (dev(a) for a in state(swag_states))
    clean = dev(synth(700, 100, 21))
    bad = clean.clone()
    bad[5, 3, 3] = float("nan")        # masked column of the v50 mask
    bad[9, 7, 12] = float("inf")       # live column
    x = clean.clone()
    idx = torch.zeros(40, dtype=torch.int32, device="cuda")
    out = torch.empty((40, 700, 2), device="cuda")
    kw = dict(philox_seed=9, single_launch=False)
    want = {"clean": ops.multiswag(clean, wa, w2, pd, idx, **kw).clone(), "bad": ops.multiswag(bad, wa, w2, pd, idx, **kw).clone()}   # also warms plan + buffers
    assert torch.isfinite(want["clean"]).all() and torch.isnan(want["bad"][:, [5, 9]]).all() and torch.isfinite(want["bad"][:, 10:]).all()
    rec = torch.full((4 + 700,), 123, dtype=torch.int32, device="cuda") if record == "caller-owned" else None

    def call():
        if rec is None:
            ops.multiswag(x, wa, w2, pd, idx, out=out, **kw)
        else:
            ops.multiswag(x, wa, w2, pd, idx, out=out, nonfinite=ops.nonfinite_scan(x, out=rec), **kw)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        call()
    same = lambda a, b: torch.equal(a.nan_to_num(nan=-7.0), b.nan_to_num(nan=-7.0)) and torch.equal(torch.isnan(a), torch.isnan(b))
    for it, kind in enumerate(("clean", "bad", "clean", "bad")):
        x.copy_(clean if kind == "clean" else bad)
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert same(out, want[kind]), (record, it, kind)
        if rec is not None:
            assert rec[:2].tolist() == ([0, 0] if kind == "clean" else [2, 1]), (it, kind, rec[:8].tolist())
    x.copy_(clean).mul_(1.01)  # new inputs in the same buffers, same graph
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ops.multiswag(x, wa, w2, pd, idx, **kw))


def test_default_route_keeps_its_scan_record_per_stream(ops, swag_states):
    """Eager calls: the default route's scan record is a module-owned buffer per (device, stream), sized on first use -- the same storage
    call after call on one stream (no allocator call per op), another buffer on another stream (concurrent streams never share a record),
    grown for a larger batch; results are those of a caller-owned record."""
    wa, w2, pd = (dev(a) for a in state(swag_states))
    x = dev(synth(300, 100, 5))
    x[7, 1, 2] = float("nan")
    idx = torch.zeros(3, dtype=torch.int32, device="cuda")
    cur = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    a = ops.multiswag(x, wa, w2, pd, idx, philox_seed=3)
    p0 = ops._records[cur].data_ptr()
    b = ops.multiswag(x, wa, w2, pd, idx, philox_seed=3)
    assert ops._records[cur].data_ptr() == p0 and torch.equal(a.nan_to_num(nan=-7.0), b.nan_to_num(nan=-7.0)) and torch.isnan(a[:, 7]).all()
    c = ops.multiswag(x, wa, w2, pd, idx, philox_seed=3, nonfinite=ops.nonfinite_scan(x))
    assert torch.equal(a.nan_to_num(nan=-7.0), c.nan_to_num(nan=-7.0))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        d = ops.multiswag(x, wa, w2, pd, idx, philox_seed=3)
        assert ops._records[(cur[0], s.cuda_stream)].data_ptr() != p0
    s.synchronize()
    assert torch.equal(a.nan_to_num(nan=-7.0), d.nan_to_num(nan=-7.0))
    big = dev(synth(5000, 100, 6))
    ops.multiswag(big, wa, w2, pd, idx, philox_seed=3)
    assert ops._records[cur].numel() >= 4 + 5000
    e = ops.multiswag(x, wa, w2, pd, idx, philox_seed=3)      # the small batch again, in the grown buffer
    assert torch.equal(a.nan_to_num(nan=-7.0), e.nan_to_num(nan=-7.0))


@pytest.mark.parametrize("B,J,nch", ((1, 1, 1), (15, 1, 1), (16, 2, 1), (17, 3, 1), (150, 20, 10), (333, 6, 3), (3000, 1, 1), (4096, 1, 1), (4097, 1, 1)))
def test_small_grids_take_the_tile_split_form_with_the_same_bits(B, J, nch, ops, swag_states):
    """Small grids -- the evaluation scripts' per-chunk calls (figures/multiswag_5_planet.py:295-298: 15-row chunks; main_figures.py:154-156:
    3 000-row batches) -- run the TILE-SPLIT launch form of the pretrained network's kernel (16 systems per workgroup, the four waves share
    a batch's tiles, wave 0 pools them in order: bnn_forward.hip.h TSPLIT).  It is a launch form, not another arithmetic: outputs, pre-clamp
    values and summaries are BIT-IDENTICAL to the plain form's (systems_per_block=64 keeps the plain form), for the in-prologue draw and the
    workspace draw, explicit and in-kernel noise, every series length the kernels take (2, 3, 5, 25 tiles: waves without a tile of their
    own), chunked draws, ragged last batches, and with damaged systems in the batch.  (4 097 rows is one workgroup too many: plain form.)"""
    wa, w2, pd = (dev(a) for a in state(swag_states))
    rng = np.random.default_rng(B * 31 + J)
    idx = torch.zeros(J, dtype=torch.int32, device="cuda")
    for T in ((100, 8, 12, 20) if B <= 333 else (100,)):
        x = dev(synth(B, T, B + T))
        for single in (True, False):
            kw = dict(nchunks=nch, philox_seed=5, single_launch=single, debug=True)
            a = ops.multiswag(x, wa, w2, pd, idx, **kw)                                  # default: tile-split when the grid is small
            b = ops.multiswag(x, wa, w2, pd, idx, systems_per_block=64, **kw)            # the plain form
            for u, v in zip(a, b):
                assert torch.equal(u, v), (B, J, nch, T, single)
        z1 = dev(rng.standard_normal((J, 7583), dtype=np.float32))
        z2 = dev(rng.standard_normal((J, 30), dtype=np.float32))
        eps = dev(rng.standard_normal((J // nch, B, 2, 20), dtype=np.float32))
        a = ops.multiswag(x, wa, w2, pd, idx, z1, z2, eps, nchunks=nch)
        assert torch.equal(a, ops.multiswag(x, wa, w2, pd, idx, z1, z2, eps, nchunks=nch, systems_per_block=64))
        W = ops.swag_draw(wa, w2, pd, idx, z1, z2)
        assert torch.equal(a, ops.forward(x, W, eps=eps, nchunks=nch)) and torch.equal(a, ops.forward(x, W, eps=eps, nchunks=nch, systems_per_block=128))
    if B >= 15:
        xb = x.clone()
        xb[3, 1, 2] = float("nan")
        xb[B - 1, 2, 20] = float("-inf")
        a = ops.multiswag(xb, wa, w2, pd, idx, nchunks=nch, philox_seed=5)
        b = ops.multiswag(xb, wa, w2, pd, idx, nchunks=nch, philox_seed=5, systems_per_block=64)
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.isnan(a[:, 3]).all() and torch.equal(a.nan_to_num(nan=-7.0), b.nan_to_num(nan=-7.0))


def test_hundred_thousand_draws_in_one_call(ops, swag_states):
    """The 'paper-ready' 5-planet setting is 10 000 samples x 10 chunks = 100 000 draws (figures/multiswag_5_planet.py:52-55,
    295-298): more than a grid.y can hold.  One call == the same call in slabs of draws, bit for bit, in both launch modes."""
    wa = dev(np.stack([swag_states[0]["w_avg"], swag_states[12]["w_avg"]]))
    w2 = dev(np.stack([swag_states[0]["w2_avg"], swag_states[12]["w2_avg"]]))
    pd = dev(np.stack([swag_states[0]["pre_D"], swag_states[12]["pre_D"]]))
    B, nch, samples = 20, 10, 10_000
    J = samples * nch
    x = dev(synth(B, 100, 3))
    idx = torch.as_tensor(np.random.default_rng(1).integers(0, 2, J).astype(np.int32))
    W = ops.swag_draw(wa, w2, pd, idx, philox_seed=21)
    assert W.shape == (J, 7583) and torch.isfinite(W).all()
    for j0 in (0, 65_530, 99_990):                                  # rows straddling the old 65 535 limit
        assert torch.equal(W[j0:j0 + 10], ops.swag_draw(wa, w2, pd, idx[j0:j0 + 10], philox_seed=21, draw_id0=j0))
    del W
    one = ops.multiswag(x, wa, w2, pd, idx, nchunks=nch, philox_seed=21, single_launch=False)
    assert one.shape == (samples, B, 2) and torch.isfinite(one).all()
    assert torch.equal(one, ops.multiswag(x, wa, w2, pd, idx, nchunks=nch, philox_seed=21, single_launch=True))
    for s0, s1 in ((0, 100), (6_500, 6_600), (9_900, 10_000)):      # slabs of samples = slabs of draws at draw_id0 = s0 * nch
        part = ops.multiswag(x, wa, w2, pd, idx[s0 * nch:s1 * nch], nchunks=nch, philox_seed=21, draw_id0=s0 * nch)
        assert torch.equal(part, one[s0:s1])


def test_two_streams_do_not_share_a_workspace(ops, swag_states):
    """Each call allocates its own draw workspace: the same op running concurrently on two streams gives each its own result."""
    wa = dev(np.stack([swag_states[0]["w_avg"], swag_states[12]["w_avg"]]))
    w2 = dev(np.stack([swag_states[0]["w2_avg"], swag_states[12]["w2_avg"]]))
    pd = dev(np.stack([swag_states[0]["pre_D"], swag_states[12]["pre_D"]]))
    x = dev(synth(2048, 100, 5))
    ia = torch.zeros(64, dtype=torch.int32)
    ib = torch.ones(64, dtype=torch.int32)
    want_a = ops.multiswag(x, wa, w2, pd, ia, philox_seed=1, single_launch=False)
    want_b = ops.multiswag(x, wa, w2, pd, ib, philox_seed=2, single_launch=False)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(sa):
            got_a = ops.multiswag(x, wa, w2, pd, ia, philox_seed=1, single_launch=False)
        with torch.cuda.stream(sb):
            got_b = ops.multiswag(x, wa, w2, pd, ib, philox_seed=2, single_launch=False)
        torch.cuda.synchronize()
        assert torch.equal(got_a, want_a) and torch.equal(got_b, want_b)


def test_mis_shaped_tensors_raise_before_any_launch(ops, swag_states):
    wa = dev(swag_states[0]["w_avg"][None]); w2 = dev(swag_states[0]["w2_avg"][None]); pd = dev(swag_states[0]["pre_D"][None])
    x = dev(synth(8, 100, 1))
    idx = torch.zeros(4, dtype=torch.int32)
    W = ops.swag_draw(wa, w2, pd, idx, philox_seed=1)
    with pytest.raises(ValueError):
        ops.forward(x, W[:, :7000].contiguous())
    with pytest.raises(ValueError):
        ops.forward(x, W[0])
    with pytest.raises(ValueError):
        ops.forward(x, W, eps=torch.zeros(4, 8, 2, 20).cuda(), eps_in=torch.zeros(4, 8, 100, 40).cuda(), eps_sum=torch.zeros(4, 8, 40).cuda())
    with pytest.raises(ValueError):
        ops.forward(x, W, eps=torch.zeros(4, 8, 2, 20).cuda(), eps_in=torch.zeros(4, 8, 100, 41).cuda())
    with pytest.raises(ValueError):
        ops.forward(x, W, nchunks=3)
    with pytest.raises(ValueError):
        ops.multiswag(x, wa[:, :100].contiguous(), w2, pd, idx)
    with pytest.raises(ValueError):
        ops.multiswag(x, wa, w2[:, :100].contiguous(), pd, idx)
    with pytest.raises(ValueError):
        ops.multiswag(x, wa, w2, pd, idx, nchunks=3)
    with pytest.raises(ValueError):
        ops.multiswag(x, wa, w2, pd, idx, z1=torch.zeros(4, 7583).cuda())
    with pytest.raises(NotImplementedError):
        ops.multiswag(x[..., :40].contiguous(), wa, w2, pd, idx)
    mom = ops.moments(torch.zeros(3, 0, 2).cuda())                  # an empty shard (more ranks than systems)
    assert mom.shape == (0, 4)


@pytest.mark.parametrize("seed", range(16))
def test_randomised_configurations_match_oracle(seed, ops, orc, swag_states):
    """A seeded sweep over the shape space -- batch size, series length, chunking, number of samples, column mask, clamp floor,
    fix_megno, block size, launch mode, two ensemble members with random picks -- each evaluated through ops.multiswag with explicit
    noise and compared with the oracle's own MC driver (orc.multiswag: its torch.chunk partition, its draw, its forward)."""
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.choice([1, 2, 5, 16, 17, 63, 64, 65, 130, 257, int(rng.integers(1, 300))]))
    T = int(rng.choice([8, 12, 36, 100, 104, 160]))
    nch_req = int(rng.choice([1, 1, 2, 3, 7, 10]))
    nch = len(torch.chunk(torch.arange(B), nch_req))
    samples = int(rng.integers(1, 4))
    megno = bool(rng.integers(0, 2))
    if rng.integers(0, 2):
        mask = ops.zero_mask_from_flags(fix_megno=megno, fix_megno2=not megno)          # the v50 column mask: 31-column kernels
    else:
        mask = int(sum(1 << int(c) for c in rng.choice(41, size=int(rng.integers(0, 12)), replace=False)))
        if megno:
            mask |= 1 << 7
    lowest = float(rng.choice([0.5, 0.1]))
    plan = ops.get_plan(mask, lowest, fix_megno=megno)
    d = plan.d
    S, K = 2, 30
    st = [swag_states[0], swag_states[12]]
    wa = np.zeros((S, d), np.float32); w2 = np.zeros((S, d), np.float32); pd = np.zeros((S, d, K), np.float32)
    for s in range(S):   # fix_megno: the v50 state stretched to d = 7665 (values are arbitrary for this test, the layout is not)
        reps = -(-d // 7583)
        wa[s] = np.tile(st[s]["w_avg"], reps)[:d]
        w2[s] = np.tile(st[s]["w2_avg"], reps)[:d]
        pd[s] = np.tile(st[s]["pre_D"], (reps, 1))[:d]
    J = samples * nch
    idx = rng.integers(0, S, J).astype(np.int32)
    z1 = rng.standard_normal((J, d), dtype=np.float32); z2 = rng.standard_normal((J, K), dtype=np.float32)
    eps = rng.standard_normal((samples, B, 2, 20), dtype=np.float32)
    x = synth(B, T, 50 + seed)
    kw = dict(systems_per_block=int(rng.choice([0, 64, 128])), single_launch=bool(rng.integers(0, 2)))
    out = ops.multiswag(dev(x), dev(wa), dev(w2), dev(pd), torch.as_tensor(idx), dev(z1), dev(z2), dev(eps), nchunks=nch, plan=plan,
                        **kw).cpu().numpy()
    arch = orc.make_arch(T=T, zero_mask=mask, lowest=lowest, fix_megno=megno)
    want = orc.multiswag(x, wa, w2, pd, idx, z1, z2, eps, nchunks=nch, arch=arch, sched=sched(ops, orc, plan))
    assert out.shape == want.shape == (samples, B, 2)
    assert np.abs(out - want).max() <= 2e-6, (seed, B, T, nch, samples, hex(mask), megno, kw, np.abs(out - want).max())
