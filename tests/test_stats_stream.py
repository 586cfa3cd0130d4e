"""The streaming statistics epilogue: truncated-normal draw + prior resampling per evaluation, fused
into the forward kernel's tail, min over trios + per-simulation quantile sketch in O(n_sims * bins) memory.
Checked against the numpy restatement in oracle/stats.py, against the un-fused path bit for bit, and against exact order
statistics of materialised samples at configs[1] size.  Needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from bnn_chaos_model_amd import ops as o
    return o


@pytest.fixture(scope="module")
def ens(swag_states):
    d = lambda k: torch.as_tensor(np.stack([swag_states[0][k], swag_states[12][k]])).cuda()
    return d("w_avg"), d("w2_avg"), d("pre_D")


def synth(B, seed):
    import bench
    return bench.synthetic_x(B, torch.device("cuda"), seed)


def test_epilogue_matches_the_oracle_restatement(ops):
    """stats_draw on arbitrary (mu, std) pairs == oracle.stats.stream_epilogue fed the very normals / uniforms the kernel draws."""
    from oracle import stats as ostats
    rng = np.random.default_rng(3)
    R, B = 7, 501
    mu = rng.uniform(4.0, 12.0, (R, B)).astype(np.float32)
    sd = rng.uniform(0.5, 6.0, (R, B)).astype(np.float32)
    mu[0, :5] = 4.0; sd[0, :5] = 0.5            # half the candidates fail
    mu[1, :5] = -40.0; sd[1, :5] = 0.5          # no candidate passes: the first one is returned (below `left`)
    musd = torch.as_tensor(np.stack([mu, sd], -1)).cuda()
    seed, row0, sys0 = 99, 12, 1_000_000_007
    st = ops.stats_params()
    got = ops.stats_draw(musd, st, philox_seed=seed, row_id0=row0, system_id0=sys0).cpu().numpy()
    cand = ops.philox_normal(5, seed, row0, R, width=40, B=B, system_id0=sys0).cpu().numpy()
    level = ops.philox_normal(6, seed, row0, R, B=B, system_id0=sys0).cpu().numpy()
    want = ostats.stream_epilogue(musd.cpu().numpy(), cand, level)
    assert np.array_equal(got, want), np.abs(got - want).max()
    assert (got[1, :5] < 4).all() and (got[2:] > 4).all()
    assert level.min() > 0 and level.max() <= 1
    # no prior resampling: the truncated-normal draw alone
    st2 = ops.stats_params(prior_threshold=None)
    got2 = ops.stats_draw(musd, st2, philox_seed=seed, row_id0=row0, system_id0=sys0).cpu().numpy()
    keep = got2 < 9
    assert np.array_equal(got2[keep], got[keep]) and (got[~keep] >= 9).all()


def test_prior_draws_follow_the_prior(ops):
    """Values redrawn from the prior have its survival function (one-sample Kolmogorov-Smirnov at the 1e-3 level)."""
    from oracle import stats as ostats
    R, B = 64, 8192
    musd = torch.zeros((R, B, 2), device="cuda")
    musd[..., 0] = 11.5
    musd[..., 1] = 0.5                                                    # nearly every draw lands above 9
    t = ops.stats_draw(musd, philox_seed=5).cpu().numpy().ravel()
    t = np.sort(t[t >= 9])
    n = t.size
    assert n > 0.99 * R * B
    S, step = ostats.prior_survival_table()
    cdf = 1.0 - np.interp(t, 9.0 + step * np.arange(S.size), S.astype(np.float64))
    emp = (np.arange(n) + 0.5) / n
    assert np.abs(cdf - emp).max() < 1.95 / np.sqrt(n)                   # K-S critical value at alpha = 0.001
    assert t.max() < 60 and abs(np.median(t) - (9 + np.log(2) / 0.424033970670719)) < 0.02


@pytest.mark.parametrize("nch", (1, 10))
def test_fused_tail_equals_forward_then_epilogue(ops, ens, nch):
    """multiswag_stats == stats_draw(multiswag): same bits; and invariant to system sharding and draw slabs."""
    wa, w2, pd = ens
    B, samples = 777, 12
    J = samples * nch
    x = synth(B, 11)
    idx = torch.as_tensor((np.arange(J) % 2).astype(np.int32))
    seed = 4242
    t = ops.multiswag_stats(x, wa, w2, pd, idx, nchunks=nch, philox_seed=seed, draw_id0=3 * nch, system_id0=50_000)
    musd = ops.multiswag(x, wa, w2, pd, idx, nchunks=nch, philox_seed=seed, draw_id0=3 * nch, system_id0=50_000)
    assert t.shape == (samples, B)
    assert torch.equal(t, ops.stats_draw(musd, philox_seed=seed, row_id0=3, system_id0=50_000))
    if nch == 1:
        part = ops.multiswag_stats(x[300:].contiguous(), wa, w2, pd, idx, philox_seed=seed, draw_id0=3, system_id0=50_300)
        assert torch.equal(part, t[:, 300:])
        part = ops.multiswag_stats(x, wa, w2, pd, idx[5:9], philox_seed=seed, draw_id0=8, system_id0=50_000)
        assert torch.equal(part, t[5:9])


def test_sketch_against_exact_percentiles_small(ops):
    """min over trios + sketch percentiles vs numpy on materialised samples: within one bin width, mean to float64 rounding."""
    rng = np.random.default_rng(8)
    R, sims, group = 1000, 257, 3
    t = rng.uniform(4.0, 9.0, (R, sims * group)).astype(np.float32)
    heavy = rng.random((R, sims * group)) < 0.3
    t[heavy] = (9.0 + rng.exponential(1 / 0.424, heavy.sum())).astype(np.float32)
    sk = ops.QuantileSketch(sims * group, group=group)
    tt = torch.as_tensor(t).cuda()
    for r0 in range(0, R, 128):                                           # slabs of draws
        sk.update(tt[r0:r0 + 128].contiguous())
    q = (2.5, 16.0, 50.0, 84.0, 97.5, 0.0, 100.0)
    got = sk.percentiles(q).cpu().numpy()
    outs = t.reshape(R, sims, group).min(2).T                             # np.min(samps_time, 2).T (:428)
    want = np.percentile(outs.astype(np.float64), q, axis=1).T
    tol = np.vectorize(sk.resolution)(want)
    assert (np.abs(got - want) <= tol * 1.0001).all(), np.abs(got - want).max()
    assert np.allclose(sk.mean().cpu().numpy(), outs.astype(np.float64).mean(1), rtol=1e-12)
    assert int(sk.hist.sum()) == R * sims and sk.count == R
    # values below the first segment (a truncated-normal draw none of whose candidates passed) are reported as its lower edge
    sk2 = ops.QuantileSketch(4)
    sk2.update(torch.tensor([[3.0, 4.5, 4.5, 4.5], [3.5, 4.5, 4.5, 4.5], [5.0, 4.5, 4.5, 4.5]]).cuda())
    got2 = sk2.percentiles((0.0, 50.0, 100.0)).cpu().numpy()
    assert got2[0, 0] == 4.0 and got2[0, 1] == 4.0 and abs(got2[0, 2] - 5.0) < 1 / 128 and abs(got2[1, 1] - 4.5) < 1 / 128
    # a NaN draw (a bad seed index poisons its draws) makes THAT simulation's percentiles and mean NaN, as np.percentile would;
    # the other simulations are untouched
    sk3 = ops.QuantileSketch(6, group=3)
    t3 = torch.tensor([[5.0, 6.0, 7.0, 5.5, 6.5, 7.5], [5.0, float("nan"), 7.0, 5.5, 6.5, 7.5], [5.0, 6.0, 7.0, 5.5, 6.5, 7.5]]).cuda()
    sk3.update(t3)
    got3 = sk3.percentiles((16.0, 50.0, 84.0)).cpu().numpy()
    assert np.isnan(got3[0]).all() and np.isnan(sk3.mean().cpu().numpy()[0])
    assert np.abs(got3[1] - 5.5).max() < 1 / 128 and abs(sk3.mean().cpu().numpy()[1] - 5.5) < 1e-12
    assert int(sk3.hist[-1].sum()) == 1 and int(sk3.hist.sum()) == 6     # the last bin is the NaN counter


def test_streaming_bands_at_configs1_size(ops, ens):
    """configs[1] size (10 000 systems x 3 000 draws): fused tail + sketch in slabs of 250 draws (never more than 10 MB of
    samples alive) vs exact order statistics of the materialised samples: every band within one bin width."""
    wa, w2, pd = ens
    B, J, slab = 10_000, 3_000, 250
    x = synth(B, 321)
    idx = torch.as_tensor((np.arange(J) % 2).astype(np.int32)).cuda()
    seed = 77
    sk = ops.QuantileSketch(B)
    for j0 in range(0, J, slab):
        sk.update(ops.multiswag_stats(x, wa, w2, pd, idx[j0:j0 + slab], philox_seed=seed, draw_id0=j0))
    q = (2.5, 16.0, 50.0, 84.0, 97.5)
    got = sk.percentiles(q)
    # exact: materialise everything (240 MB of pairs), epilogue, per-system percentiles by sorting
    musd = ops.multiswag(x, wa, w2, pd, idx, philox_seed=seed)
    t = ops.stats_draw(musd, philox_seed=seed)
    want = torch.quantile(t.double(), torch.tensor(q, dtype=torch.float64, device="cuda") / 100.0, dim=0).T   # 'linear', as numpy
    exact_kernel = ops.quantiles(torch.stack([t, t], 2).contiguous(), q)[:, 0, :]
    assert torch.allclose(exact_kernel.double(), want, rtol=0, atol=1e-6)
    err = (got.double() - want).abs().cpu().numpy()
    tol = np.vectorize(sk.resolution)(want.cpu().numpy())
    assert (err <= tol * 1.0001).all(), (err.max(), err.argmax())
    assert np.median(err) < 0.004                                          # typically well inside a bin
    assert torch.allclose(sk.mean(), t.double().mean(0), rtol=1e-12)
    assert sk.hist.numel() * 4 + sk.mom.numel() * 8 < 40e6                # O(B * bins): 38 MB, whatever the number of draws


def test_sharded_driver_bands_single_rank(ops, ens):
    """MultiSwagSharded.predictive_quantiles (the 5-planet shape: trios of systems, min over the trio) at world size 1 ==
    exact percentiles of the materialised post-epilogue samples within one bin width; the gather is a no-op here (the N > 1
    gather is covered on CPU ranks by tests/test_host_cpu.py::test_sharded_bands_gather_gloo)."""
    from bnn_chaos_model_amd.distributed import MultiSwagSharded
    wa, w2, pd = ens
    sims, J = 500, 600
    x = synth(sims * 3, 9)
    idx = torch.as_tensor((np.arange(J) % 2).astype(np.int32)).cuda()
    drv = MultiSwagSharded(wa, w2, pd, draws_per_launch=128)
    q = (2.5, 16.0, 50.0, 84.0, 97.5)
    got = drv.predictive_quantiles(x, sims * 3, idx, q=q, philox_seed=13, trios=3)
    assert got.shape == (sims, len(q) + 1)
    t = ops.stats_draw(ops.multiswag(x, wa, w2, pd, idx, philox_seed=13), philox_seed=13)        # [J, sims*3]
    outs = t.reshape(J, sims, 3).min(2).values.double()                                                  # np.min(samps_time, 2)
    want = torch.quantile(outs, torch.tensor(q, dtype=torch.float64, device="cuda") / 100.0, dim=0).T
    sk = ops.QuantileSketch(3, group=3)
    tol = np.vectorize(sk.resolution)(want.cpu().numpy())
    err = (got[:, :len(q)].double() - want).abs().cpu().numpy()
    assert (err <= tol * 1.0001).all(), err.max()
    assert torch.allclose(got[:, -1].double(), outs.mean(0), rtol=1e-6)
    with pytest.raises(ValueError):
        drv.predictive_quantiles(x[:-1], sims * 3, idx, trios=3)


def test_native_slab_drivers(ops, ens):
    """bnn_multiswag_moments_f64 / bnn_multiswag_bands_f32 (the slab loops inside the library) == the same slabs driven from
    Python, for a dense grid and for chunked draws; empty inputs are no-ops."""
    wa, w2, pd = ens
    B, nch, samples = 999, 3, 40
    J = samples * nch
    x = synth(B, 4)
    idx = torch.as_tensor((np.arange(J) % 2).astype(np.int32)).cuda()
    mom = ops.multiswag_moments(x, wa, w2, pd, idx, nchunks=nch, philox_seed=5, draws_per_launch=33, system_id0=10)   # slabs of 33 -> 33 draws
    want = ops.moments(ops.multiswag(x, wa, w2, pd, idx, nchunks=nch, philox_seed=5, system_id0=10))
    assert torch.allclose(mom, want, rtol=1e-13, atol=0)
    sk = ops.QuantileSketch(B, group=3)
    ops.multiswag_bands(x, wa, w2, pd, idx, sk, nchunks=nch, philox_seed=5, draws_per_launch=30, system_id0=10)
    sk2 = ops.QuantileSketch(B, group=3)
    sk2.update(ops.multiswag_stats(x, wa, w2, pd, idx, nchunks=nch, philox_seed=5, system_id0=10))
    assert torch.equal(sk.hist, sk2.hist) and sk.count == samples == sk2.count
    assert torch.allclose(sk.mom, sk2.mom, rtol=1e-13, atol=0)
    # empty shard / no draws
    e = ops.multiswag_moments(x[:0], wa, w2, pd, idx, nchunks=nch, philox_seed=5)
    assert e.shape == (0, 4)
    z = ops.multiswag_moments(x, wa, w2, pd, idx[:0], philox_seed=5)
    assert z.shape == (B, 4) and (z == 0).all()
    with pytest.raises(ValueError):
        ops.multiswag_bands(x, wa, w2, pd, idx, ops.QuantileSketch(B - 3, group=3))


def test_streamed_bands_agree_with_the_reference_replay_statistically(ops, ens):
    """The link between the STREAMED epilogue (Philox candidates, exact erfc survival table, quantile sketch) and the reference's
    own arithmetic (the numpy-replay kernels of bnn_chaos_model_amd/stats.py, bit-identical to the reference's source fragments,
    figures/multiswag_5_planet.py:388-428, 484-489): on the SAME [R, B, 2] (mu, std) samples, R = 4000 draws of 100 simulations x 3
    trios, the two pipelines are independent Monte-Carlo estimates of the same per-simulation distribution, so every band must
    agree within   sketch bin width  +  z = 5 sigma of the order-statistic sampling error of the DIFFERENCE of two estimates.
    The sampling error is taken distribution-free from the replay sample itself: the q-quantile of R draws lies between the
    order statistics of ranks R q -+ z sqrt(2 R q (1 - q)) (binomial), so the bound is that rank interval's width."""
    from scipy import stats as sst
    from bnn_chaos_model_amd import stats
    wa, w2, pd = ens
    R, sims, trios = 4000, 100, 3
    B = sims * trios
    x = synth(B, 2718)
    idx = torch.as_tensor((np.arange(R) % 2).astype(np.int32)).cuda()
    musd = ops.multiswag(x, wa, w2, pd, idx, philox_seed=31)                       # [R, B, 2]
    q = (2.5, 16.0, 50.0, 84.0, 97.5)
    # (a) the reference's pipeline, numpy generator consumed exactly as the script does
    np.random.seed(12345)
    samps = stats.fast_truncnorm(musd, left=4, nsamp=40, d=10000)                  # :388-392
    samps = stats.resample_prior(samps)                                            # :396-422
    outs = stats.min_over_trios(samps.reshape(R, sims, trios))                     # :428 -> [sims, R]
    ref = stats.percentiles(outs, q).double().cpu().numpy()                        # :484-489
    ref_mean = outs.double().mean(1).cpu().numpy()
    # (b) the streamed pipeline
    t = ops.stats_draw(musd, philox_seed=77)
    sk = ops.QuantileSketch(B, group=trios)
    for r0 in range(0, R, 500):
        sk.update(t[r0:r0 + 500].contiguous())
    got = sk.percentiles(q).double().cpu().numpy()
    got_mean = sk.mean().cpu().numpy()
    # bound per (simulation, band)
    z = 5.0
    srt = np.sort(outs.double().cpu().numpy(), axis=1)                             # [sims, R]
    worst = 0.0
    for k, qq in enumerate(q):
        p = qq / 100.0
        half = z * np.sqrt(2.0 * R * p * (1 - p))
        lo = np.clip(np.floor(R * p - half).astype(int), 0, R - 1)
        hi = np.clip(np.ceil(R * p + half).astype(int), 0, R - 1)
        bound = (srt[:, hi] - srt[:, lo]) + np.vectorize(sk.resolution)(ref[:, k])
        err = np.abs(got[:, k] - ref[:, k])
        worst = max(worst, (err / bound).max())
        assert (err <= bound).all(), (qq, err.max(), bound[err.argmax()])
    sd = srt.std(1)
    assert (np.abs(got_mean - ref_mean) <= z * np.sqrt(2.0) * sd / np.sqrt(R)).all()
    # and element-wise: the two samples of post-epilogue times have the same distribution on both legs (two-sample K-S, alpha = 1e-3)
    a = samps.cpu().numpy().ravel()[::3]       # the same (draw, system) elements in both: each pair of entries is two independent
    b = t.cpu().numpy().ravel()[::3]           # draws from the same conditional distribution, so the pooled samples match in law
    for leg in (lambda v: v[v < 9], lambda v: v[v >= 9]):
        va, vb = leg(a), leg(b)
        assert min(va.size, vb.size) > 10_000
        ks = sst.ks_2samp(va, vb)
        assert ks.statistic < 1.95 * np.sqrt((va.size + vb.size) / (va.size * vb.size)), ks
    assert abs((a >= 9).mean() - (b >= 9).mean()) < 5 * np.sqrt(0.5 / a.size)       # the same share of draws reaches the prior
