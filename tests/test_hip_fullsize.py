"""BASELINE.json configs[1] at FULL size (10 000 systems x 30 seeds x 100 samples = 3e7 evals) through
size-independent properties, plus oracle spot checks on randomly chosen (draw, system) pairs.  Needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, S, M = 10_000, 30, 100
J = S * M
SEED = 20240


@pytest.fixture(scope="module")
def full():
    import bench
    from bnn_chaos_model_amd import ops
    dev = torch.device("cuda")
    x = bench.synthetic_x(B, dev, 321)
    wa, w2, pd = bench.synthetic_ensemble(S, dev)
    idx = (torch.arange(J, dtype=torch.int32) % S).to(dev)
    out = ops.multiswag(x, wa, w2, pd, idx, philox_seed=SEED)
    torch.cuda.synchronize()
    return dict(ops=ops, x=x, wa=wa, w2=w2, pd=pd, idx=idx, out=out)


def test_ranges_and_finiteness(full):
    out = full["out"]
    assert out.shape == (J, B, 2) and torch.isfinite(out).all()
    mu, sd = out[..., 0], out[..., 1]
    assert mu.min() >= 4 and mu.max() <= 12 and sd.min() >= 0.5 and sd.max() <= 6
    assert 4.5 < mu.mean().item() < 11 and mu.std().item() > 0.5  # the synthetic "slow" inputs spread mu over the range


def test_system_sharding_is_bit_invariant(full):
    """Evaluating a slice of the systems with its global offset reproduces the corresponding slice."""
    o, f = full["ops"], full
    for lo, hi in ((0, 1250), (1250, 2500), (8750, 10000), (4999, 5003)):
        part = o.multiswag(f["x"][lo:hi].contiguous(), f["wa"], f["w2"], f["pd"], f["idx"], philox_seed=SEED, system_id0=lo)
        assert torch.equal(part, f["out"][:, lo:hi])


def test_draw_slabs_are_bit_invariant(full):
    """Draws evaluated in slabs with draw_id0 (as distributed.MultiSwagSharded does) reproduce the one-launch result."""
    o, f = full["ops"], full
    for j0, j1 in ((0, 256), (256, 512), (2900, 3000)):
        part = o.multiswag(f["x"], f["wa"], f["w2"], f["pd"], f["idx"][j0:j1], philox_seed=SEED, draw_id0=j0)
        assert torch.equal(part, f["out"][j0:j1])


def test_block_size_and_launch_mode_do_not_change_bits(full):
    o, f = full["ops"], full
    sub = slice(0, 300)
    ref = f["out"][sub]
    for kw in (dict(systems_per_block=64), dict(systems_per_block=256), dict(single_launch=True), dict(single_launch=True, systems_per_block=128)):
        got = o.multiswag(f["x"], f["wa"], f["w2"], f["pd"], f["idx"][sub], philox_seed=SEED, **kw)
        assert torch.equal(got, ref), kw


def test_masked_columns_and_time_order_do_not_matter(full):
    o, f = full["ops"], full
    sub = slice(0, 60)
    x2 = f["x"].clone()
    x2[:, :, [1, 2, 3, 4, 5, 6, 7, 38, 39, 40]] = 1e6
    assert torch.equal(o.multiswag(x2, f["wa"], f["w2"], f["pd"], f["idx"][sub], philox_seed=SEED), f["out"][sub])
    perm = torch.randperm(100, generator=torch.Generator().manual_seed(0)).cuda()
    got = o.multiswag(f["x"][:, perm].contiguous(), f["wa"], f["w2"], f["pd"], f["idx"][sub], philox_seed=SEED)
    err = (got - f["out"][sub]).abs()
    assert (err <= 1e-5 + 1e-5 * f["out"][sub].abs()).all()  # the time pool is order invariant up to fp32 rounding


def test_moments_match_a_float64_reduction(full):
    o, f = full["ops"], full
    mom = o.moments(f["out"])
    mu = f["out"][..., 0].double()
    sd = f["out"][..., 1].double()
    want = torch.stack([mu.sum(0), (mu * mu).sum(0), sd.sum(0), (sd * sd).sum(0)], 1)
    assert torch.allclose(mom, want, rtol=1e-12, atol=0)


def test_oracle_spot_checks_at_full_size(full):
    """64 random (draw, system) pairs of the 3e7 evaluated against the CPU oracle, fed the very normals the kernel generated."""
    from oracle import oracle as orc
    o, f = full["ops"], full
    rng = np.random.default_rng(5)
    draws = np.unique(rng.integers(0, J, 8))
    systems = np.unique(rng.integers(0, B, 8))
    plan = o.get_plan()
    sched = orc.make_schedule([plan.layer_order(l) for l in range(6)], pool_parts=4)
    wa, w2, pd = (t.cpu().numpy() for t in (f["wa"], f["w2"], f["pd"]))
    z1 = o.philox_normal(0, SEED, 0, J, width=7583).cpu().numpy()
    z2 = o.philox_normal(1, SEED, 0, J, width=30).cpu().numpy()
    worst = 0.0
    for j in draws:
        s = int(f["idx"][j])
        w = orc.swag_draw(wa[s], w2[s], pd[s], z1[j], z2[j])
        for b in systems:
            eps = o.philox_normal(2, SEED, int(j), 1, B=1, system_id0=int(b)).cpu().numpy()[0, 0]
            ref = orc.forward(f["x"][b:b + 1].cpu().numpy(), w, eps[0:1], eps[1:2], sched=sched)[0]
            got = f["out"][j, b].cpu().numpy()
            worst = max(worst, np.abs(got - ref).max())
    assert worst <= 2e-6, worst


def test_multiswag_sharded_driver_single_rank(full):
    """distributed.MultiSwagSharded (draws in slabs, moments accumulated, gather a no-op at world size 1) == one launch."""
    from bnn_chaos_model_amd.distributed import MultiSwagSharded, moments_to_mean_std
    f = full
    drv = MultiSwagSharded(f["wa"], f["w2"], f["pd"], draws_per_launch=256)
    sub = slice(0, 2000)
    mom = drv.predictive_moments(f["x"][sub].contiguous(), 2000, f["idx"][:700], philox_seed=SEED)
    want = f["ops"].moments(f["out"][:700, sub].contiguous())
    assert torch.allclose(mom, want, rtol=1e-13, atol=0)  # slabs of 256 draws vs one launch: float64 sums, different association
    st = moments_to_mean_std(mom, 700)
    assert st["mean_mu"].min() >= 4 and st["mean_mu"].max() <= 12 and (st["std_mu"] >= 0).all()


# ---- BASELINE.json configs[2] at FULL size: 1 000 000 systems x 100 draws = 1e8 evals; x has 4.1e9 elements (> 2^31) ----------
B3, J3 = 1_000_000, 100


@pytest.fixture(scope="module")
def full_c3():
    import bench
    from bnn_chaos_model_amd import ops
    dev = torch.device("cuda")
    x = bench.synthetic_x(B3, dev, 777)
    assert x.numel() > 2 ** 31
    wa, w2, pd = bench.synthetic_ensemble(S, dev)
    idx = (torch.arange(J3, dtype=torch.int32) % S).to(dev)
    out = ops.multiswag(x, wa, w2, pd, idx, philox_seed=SEED + 1)
    torch.cuda.synchronize()
    yield dict(ops=ops, x=x, wa=wa, w2=w2, pd=pd, idx=idx, out=out)
    del x, out
    torch.cuda.empty_cache()


def test_configs2_oracle_spot_checks_beyond_2_31_elements(full_c3):
    """64 (draw, system) pairs of the 1e8 against the CPU oracle fed the very normals the kernel generated; the systems include
    the first, the ones either side of element 2^31 of x, and the last two."""
    from oracle import oracle as orc
    o, f = full_c3["ops"], full_c3
    out = f["out"]
    assert out.shape == (J3, B3, 2) and torch.isfinite(out).all()
    mu, sd = out[..., 0], out[..., 1]
    assert mu.min() >= 4 and mu.max() <= 12 and sd.min() >= 0.5 and sd.max() <= 6
    edge = 2 ** 31 // 4100                                      # system whose row holds element 2^31
    rng = np.random.default_rng(6)
    systems = [0, edge - 1, edge, edge + 1, B3 - 2, B3 - 1] + rng.integers(edge, B3, 2).tolist()
    draws = [0, J3 - 1] + rng.integers(1, J3 - 1, 6).tolist()
    plan = o.get_plan()
    sched = orc.make_schedule([plan.layer_order(l) for l in range(6)], pool_parts=4)
    wa, w2, pd = (t.cpu().numpy() for t in (f["wa"], f["w2"], f["pd"]))
    z1 = o.philox_normal(0, SEED + 1, 0, J3, width=7583).cpu().numpy()
    z2 = o.philox_normal(1, SEED + 1, 0, J3, width=30).cpu().numpy()
    worst, n = 0.0, 0
    for j in draws:
        s = int(f["idx"][j])
        w = orc.swag_draw(wa[s], w2[s], pd[s], z1[j], z2[j])
        for b in systems:
            eps = o.philox_normal(2, SEED + 1, int(j), 1, B=1, system_id0=int(b)).cpu().numpy()[0, 0]
            ref = orc.forward(f["x"][b:b + 1].cpu().numpy(), w, eps[0:1], eps[1:2], sched=sched)[0]
            worst = max(worst, np.abs(out[j, b].cpu().numpy() - ref).max())
            n += 1
    assert n == 64 and worst <= 2e-6, worst


def test_configs2_sharding_and_launch_mode_invariance(full_c3):
    """A slice that starts beyond element 2^31, evaluated on its own with its global offset, and the in-prologue-draw mode on a
    slab of draws reproduce the one-launch result bit for bit."""
    o, f = full_c3["ops"], full_c3
    lo, hi = 900_000, 900_000 + 4096
    part = o.multiswag(f["x"][lo:hi].contiguous(), f["wa"], f["w2"], f["pd"], f["idx"], philox_seed=SEED + 1, system_id0=lo)
    assert torch.equal(part, f["out"][:, lo:hi])
    got = o.multiswag(f["x"], f["wa"], f["w2"], f["pd"], f["idx"][40:44], philox_seed=SEED + 1, draw_id0=40, single_launch=True)
    assert torch.equal(got, f["out"][40:44])
    mom = o.moments(f["out"])
    want = f["out"][:, -1000:, 0].double().sum(0)
    assert torch.allclose(mom[-1000:, 0], want, rtol=1e-12, atol=0)


def test_noisy_forward_oracle_spot_checks(full):
    """forward(noisy_val=True) with every normal generated in-kernel, 10 000 systems x 64 draws: 48 (draw, system) pairs against
    the CPU oracle fed the very normals the kernel generated (input noise: six normals per Philox block; pool and summary noise)."""
    from oracle import oracle as orc
    o, f = full["ops"], full
    Jn = 64
    W = o.swag_draw(f["wa"], f["w2"], f["pd"], f["idx"][:Jn], philox_seed=SEED + 2)
    out = o.forward(f["x"], W, philox_seed=SEED + 2, draw_id0=5, system_id0=123_456, noisy=True)
    assert out.shape == (Jn, B, 2) and torch.isfinite(out).all()
    rng = np.random.default_rng(9)
    draws = [0, Jn - 1] + rng.integers(1, Jn - 1, 4).tolist()
    systems = [0, B - 1] + rng.integers(1, B - 1, 6).tolist()
    plan = o.get_plan()
    sched = orc.make_schedule([plan.layer_order(l, noisy=True) for l in range(6)], pool_parts=4)
    Wc = W.cpu().numpy()
    worst = 0.0
    for j in draws:
        for b in systems:
            kw = dict(B=1, system_id0=123_456 + int(b))
            eps = o.philox_normal(2, SEED + 2, 5 + int(j), 1, **kw).cpu().numpy()[0, 0]
            e_in = o.philox_normal(3, SEED + 2, 5 + int(j), 1, width=100, **kw).cpu().numpy()[0]
            e_sum = o.philox_normal(4, SEED + 2, 5 + int(j), 1, **kw).cpu().numpy()[0]
            ref = orc.forward(f["x"][b:b + 1].cpu().numpy(), Wc[j], eps[0:1], eps[1:2], eps_in=e_in, eps_sum=e_sum, sched=sched)[0]
            worst = max(worst, np.abs(out[j, b].cpu().numpy() - ref).max())
    assert worst <= 2e-6, worst
