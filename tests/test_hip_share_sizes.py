"""One GPU's share of the two 8-GPU configurations of BASELINE.json at FULL per-GPU size, with the GLOBAL offsets the last rank
of an 8-GPU job carries, against the CPU oracle fed the very normals the kernel generated:

  configs[3]: 10M systems x 3000 draws over 8 GPUs -> 1.25M systems per GPU (x = 20.5 GB, the largest x this code ever sees);
              rank 7 owns global systems [8 750 000, 10 000 000); its last slab of draws is [2750, 3000).
  configs[4]: x [1e6, 3, 100, 41] over 8 GPUs -> 125 000 simulations = 375 000 rows per GPU, 10 chunks (torch.chunk) x samples,
              one (member, draw) per chunk per sample, XCD-aware work order on; rank 7 owns rows [2 625 000, 3 000 000).

Needs an MI355X (about 25 GB of HBM)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

S = 30
SEED = 31337


def big_x(B, seed0, piece=125_000):
    import bench
    dev = torch.device("cuda")
    x = torch.empty((B, 100, 41), dtype=torch.float32, device=dev)
    for i, lo in enumerate(range(0, B, piece)):
        hi = min(B, lo + piece)
        x[lo:hi] = bench.synthetic_x(hi - lo, dev, seed0 + i)
    return x


# ---- configs[3] share ---------------------------------------------------------------------------------------------------------
B4, SYS0_4, DRAW0_4, NSLAB = 1_250_000, 8_750_000, 2750, 250


@pytest.fixture(scope="module")
def c4_share():
    import bench
    from bnn_chaos_model_amd import ops
    dev = torch.device("cuda")
    x = big_x(B4, 4000)
    assert x.numel() * 4 > 20e9
    wa, w2, pd = bench.synthetic_ensemble(S, dev)
    idx_all = (torch.arange(3000, dtype=torch.int32) % S).to(dev)
    idx = idx_all[DRAW0_4:DRAW0_4 + NSLAB].contiguous()
    out = ops.multiswag(x, wa, w2, pd, idx, philox_seed=SEED, draw_id0=DRAW0_4, system_id0=SYS0_4)
    torch.cuda.synchronize()
    yield dict(ops=ops, x=x, wa=wa, w2=w2, pd=pd, idx=idx, out=out)
    del x, out
    torch.cuda.empty_cache()


def test_configs3_share_oracle_spot_checks_with_rank7_offsets(c4_share):
    """1.25M systems x rank 7's last slab of 250 draws (system_id0 = 8 750 000, draw_id0 = 2750): 64 (draw, system) pairs --
    the first and last two systems, the first and last draw, random ones -- against the oracle."""
    from oracle import oracle as orc
    o, f = c4_share["ops"], c4_share
    out = f["out"]
    assert out.shape == (NSLAB, B4, 2) and torch.isfinite(out).all()
    mu, sd = out[..., 0], out[..., 1]
    assert mu.min() >= 4 and mu.max() <= 12 and sd.min() >= 0.5 and sd.max() <= 6
    rng = np.random.default_rng(11)
    systems = [0, 1, B4 - 2, B4 - 1] + rng.integers(2, B4 - 2, 4).tolist()
    draws = [0, NSLAB - 1] + rng.integers(1, NSLAB - 1, 6).tolist()
    plan = o.get_plan()
    sched = orc.make_schedule([plan.layer_order(l) for l in range(6)], pool_parts=4)
    wa, w2, pd = (t.cpu().numpy() for t in (f["wa"], f["w2"], f["pd"]))
    z1 = o.philox_normal(0, SEED, DRAW0_4, NSLAB, width=7583).cpu().numpy()
    z2 = o.philox_normal(1, SEED, DRAW0_4, NSLAB, width=30).cpu().numpy()
    worst, n = 0.0, 0
    for j in draws:
        s = int(f["idx"][j])
        assert s == (DRAW0_4 + j) % S
        w = orc.swag_draw(wa[s], w2[s], pd[s], z1[j], z2[j])
        for b in systems:
            eps = o.philox_normal(2, SEED, DRAW0_4 + int(j), 1, B=1, system_id0=SYS0_4 + int(b)).cpu().numpy()[0, 0]
            ref = orc.forward(f["x"][b:b + 1].cpu().numpy(), w, eps[0:1], eps[1:2], sched=sched)[0]
            worst = max(worst, np.abs(out[j, b].cpu().numpy() - ref).max())
            n += 1
    assert n == 64 and worst <= 2e-6, worst


def test_configs3_share_slab_driver_equals_the_materialised_slab(c4_share):
    """The native slab driver (bnn_multiswag_moments_f64, what MultiSwagSharded and `bench.py --workload c4` run) on the same slab
    in two sub-slabs of 125 draws == float64 moments of the materialised samples; and a shard evaluated on its own reproduces
    its slice bit for bit (global ids, not launch geometry, key the noise)."""
    o, f = c4_share["ops"], c4_share
    mom = o.multiswag_moments(f["x"], f["wa"], f["w2"], f["pd"], f["idx"], philox_seed=SEED, draw_id0=DRAW0_4, system_id0=SYS0_4,
                              draws_per_launch=125)
    want = o.moments(f["out"])
    assert torch.allclose(mom, want, rtol=1e-13, atol=0)
    lo, hi = B4 - 5000, B4
    part = o.multiswag(f["x"][lo:hi].contiguous(), f["wa"], f["w2"], f["pd"], f["idx"][100:104], philox_seed=SEED, draw_id0=DRAW0_4 + 100,
                       system_id0=SYS0_4 + lo)
    assert torch.equal(part, f["out"][100:104, lo:hi])


# ---- configs[4] share ---------------------------------------------------------------------------------------------------------
B5, NCH, SAMPLES, SYS0_5, ROW0_5 = 375_000, 10, 2, 2_625_000, 40


@pytest.fixture(scope="module")
def c5_share():
    import bench
    from bnn_chaos_model_amd import ops
    dev = torch.device("cuda")
    x = big_x(B5, 5000)
    wa, w2, pd = bench.synthetic_ensemble(S, dev)
    idx = torch.as_tensor(np.random.default_rng(3).integers(0, S, SAMPLES * NCH).astype(np.int32)).to(dev)
    yield dict(ops=ops, x=x, wa=wa, w2=w2, pd=pd, idx=idx)
    del x
    torch.cuda.empty_cache()


def chunk_edges(B, nch):
    csz = -(-B // nch)
    return [(c, c * csz, min(B, (c + 1) * csz)) for c in range(nch)]


def test_configs4_share_fp32_oracle_at_every_chunk_edge(c5_share):
    """375k rows x 10 chunks x 2 samples, rank 7's offsets (system_id0 = 2 625 000, draw_id0 = 400 = output row 40): the first and
    last row of EVERY chunk under both samples (40 pairs; the grid is chunked, so the XCD-aware work order is on) against the
    oracle; the output row r of chunk c uses draw r * 10 + c."""
    from oracle import oracle as orc
    o, f = c5_share["ops"], c5_share
    draw0 = ROW0_5 * NCH
    out = o.multiswag(f["x"], f["wa"], f["w2"], f["pd"], f["idx"], nchunks=NCH, philox_seed=SEED + 1, draw_id0=draw0, system_id0=SYS0_5)
    assert out.shape == (SAMPLES, B5, 2) and torch.isfinite(out).all()
    plan = o.get_plan()
    sched = orc.make_schedule([plan.layer_order(l) for l in range(6)], pool_parts=4)
    wa, w2, pd = (t.cpu().numpy() for t in (f["wa"], f["w2"], f["pd"]))
    J = SAMPLES * NCH
    z1 = o.philox_normal(0, SEED + 1, draw0, J, width=7583).cpu().numpy()
    z2 = o.philox_normal(1, SEED + 1, draw0, J, width=30).cpu().numpy()
    idx = f["idx"].cpu().numpy()
    worst, n = 0.0, 0
    for r in range(SAMPLES):
        for c, lo, hi in chunk_edges(B5, NCH):
            e = r * NCH + c
            w = orc.swag_draw(wa[idx[e]], w2[idx[e]], pd[idx[e]], z1[e], z2[e])
            for b in (lo, hi - 1):
                eps = o.philox_normal(2, SEED + 1, ROW0_5 + r, 1, B=1, system_id0=SYS0_5 + b).cpu().numpy()[0, 0]
                ref = orc.forward(f["x"][b:b + 1].cpu().numpy(), w, eps[0:1], eps[1:2], sched=sched)[0]
                worst = max(worst, np.abs(out[r, b].cpu().numpy() - ref).max())
                n += 1
    assert n == 40 and worst <= 2e-6, worst
    # the two launch modes and a different block size give the same bits on the chunked grid
    again = o.multiswag(f["x"], f["wa"], f["w2"], f["pd"], f["idx"], nchunks=NCH, philox_seed=SEED + 1, draw_id0=draw0, system_id0=SYS0_5,
                        single_launch=True, systems_per_block=128)
    assert torch.equal(again, out)


def test_configs4_share_f16x3_emulation_at_every_chunk_edge(c5_share):
    """The opt-in f16x3 form on the same chunked grid: pooled summaries (pool noise zero) of the first and last row of every chunk
    against the float64 emulation of the same operand splitting (oracle/lowp.py), same bound as tests/test_lowp.py (2e-5 of the
    row scale); and its (mu, std) stay within 1e-4 of the fp32 path on those rows."""
    from oracle import lowp
    o, f = c5_share["ops"], c5_share
    draw0 = ROW0_5 * NCH
    W = o.swag_draw(f["wa"], f["w2"], f["pd"], f["idx"], philox_seed=SEED + 1, draw_id0=draw0)
    eps = torch.zeros((SAMPLES, B5, 2, 20), dtype=torch.float32, device="cuda")
    out, pre, summ = o.forward(f["x"], W, eps=eps, nchunks=NCH, draw_id0=draw0, system_id0=SYS0_5, debug=True, precision="f16x3")
    o32 = o.forward(f["x"], W, eps=eps, nchunks=NCH, draw_id0=draw0, system_id0=SYS0_5)
    Wc = W.cpu().numpy()
    worst, n = 0.0, 0
    for r in range(SAMPLES):
        for c, lo, hi in chunk_edges(B5, NCH):
            e = r * NCH + c
            for b in (lo, hi - 1):
                xb = f["x"][b:b + 1].cpu().numpy()
                want = lowp.pooled_summary(lowp.feature_nn(xb, Wc[e], 2, "f16"))[0]
                got = summ[r, b].cpu().numpy().astype(np.float64)
                worst = max(worst, (np.abs(got - want) / np.abs(want).max()).max())
                assert (out[r, b] - o32[r, b]).abs().max().item() < 1e-4
                n += 1
    assert n == 40 and worst <= 2e-5, worst


def test_configs3_share_streamed_bands_equal_exact_order_statistics(c4_share):
    """The streamed form at share size (what `bench.py --workload c4q` and MultiSwagSharded.local_bands run): the native slab driver
    with the fused statistics tail + quantile sketch over all 1.25M systems x the 250-draw slab, against exact percentiles of the
    post-epilogue times computed from the materialised samples for the first and the last 2000 systems -- within one bin width."""
    o, f = c4_share["ops"], c4_share
    q = (2.5, 16.0, 50.0, 84.0, 97.5)
    sk = o.QuantileSketch(B4)
    o.multiswag_bands(f["x"], f["wa"], f["w2"], f["pd"], f["idx"], sk, philox_seed=SEED, draw_id0=DRAW0_4, system_id0=SYS0_4, draws_per_launch=125)
    got = sk.percentiles(q)
    assert got.shape == (B4, 5) and torch.isfinite(got).all() and sk.count == NSLAB
    for lo, hi in ((0, 2000), (B4 - 2000, B4)):
        t = o.stats_draw(f["out"][:, lo:hi].contiguous(), philox_seed=SEED, row_id0=DRAW0_4, system_id0=SYS0_4 + lo)   # [250, 2000]
        want = torch.quantile(t.double(), torch.tensor(q, dtype=torch.float64, device="cuda") / 100.0, dim=0).T
        err = (got[lo:hi].double() - want).abs().cpu().numpy()
        # a percentile is interpolated between two neighbouring order statistics, each of which the sketch places inside its own bin:
        # the bound is the width of the coarser of those two bins (with only 250 draws the tail percentiles often straddle a segment
        # boundary of the 1/128 | 1/32 | 1/2 wide bins)
        srt = torch.sort(t.double(), dim=0).values.cpu().numpy()                      # [250, 2000]
        res = np.vectorize(sk.resolution)
        tol = np.empty_like(err)
        for k, qq in enumerate(q):
            r0 = int(np.floor(qq / 100.0 * (NSLAB - 1)))
            r1 = min(r0 + 1, NSLAB - 1)
            tol[:, k] = np.maximum(res(srt[r0]), res(srt[r1]))
        assert (err <= tol * 1.0001).all(), (lo, err.max())
        assert np.median(err) < 0.004
        assert torch.allclose(sk.mean()[lo:hi], t.double().mean(0), rtol=1e-12)
