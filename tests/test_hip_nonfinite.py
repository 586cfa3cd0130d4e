"""NON-FINITE inputs on the GPU: NaN exactly where the unmodified reference has NaN, 1e-5 where it is finite
(tests/golden/make_golden_nonfinite.py -> case_nonfinite.npz; reference spock_reg_model.py:452-478 `x = x - mask`, :301-321 nn.ReLU,
:416-435).  Covers the scan (which systems, which are certain), the exact re-evaluation behind every forward entry point -- both
engines, fused and workspace draws, quiet and noisy, explicit and in-kernel noise, chunked draws, sharded batches, the statistics tail,
the slab drivers, the side-effect outputs -- the `dead` network whose +inf dies in a ReLU (finite outputs), and the module surface.
Needs an MI355X."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

CERTAIN = {1, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 17, 18}      # NaN anywhere, or +-inf in a masked column
EXACT = {2, 3, 13, 16}                                            # +-inf in live columns only: needs the evaluation


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def tp(z, pfx):
    return [z[f"{pfx}_{i:03d}"] for i in range(int(z[pfx + "_n"]))]


def same_nan_close_elsewhere(got, want, rtol=1e-5):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want)), np.argwhere(np.isnan(got) != np.isnan(want))[:8]
    inf = np.isinf(want)
    assert np.array_equal(got[inf], want[inf])
    fin = np.isfinite(want)
    err = np.abs(got[fin] - want[fin])
    assert (err <= rtol * np.abs(want[fin])).all(), err.max()


@pytest.fixture(scope="module")
def ops():
    from bnn_chaos_model_amd import ops as o
    return o


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def z():
    return load_golden("case_nonfinite.npz")


def state(swag_states, si):
    st = swag_states[si]
    return dev(st["w_avg"][None]), dev(st["w2_avg"][None]), dev(st["pre_D"][None])


def dead_plan(ops, orc, z):
    hp = json.loads(str(z["dead_hparams_json"]))
    kw = dict(n_features=41, hidden=int(hp["hidden"]), latent=int(hp["latent"]), depth_in=int(hp["in"]), depth_out=int(hp["out"]))
    return ops.get_plan(ops.V50_ZERO_MASK, 0.5, **kw), orc.make_arch(T=100, **{k: v for k, v in kw.items() if k != "n_features"})


def test_scan_lists_the_damaged_systems_and_knows_which_are_certain(ops, z):
    x = dev(z["x"])
    rec = ops.nonfinite_scan(x).cpu().numpy()
    n = int(rec[0])
    ent = rec[4:4 + n]
    assert sorted(ent >> 1) == sorted(CERTAIN | EXACT) and int(rec[1]) == len(CERTAIN)
    assert {int(e >> 1) for e in ent if e & 1} == CERTAIN
    # a clean batch: an empty list; an empty batch: an empty list; a ragged series length and an unaligned view: the same answers
    assert int(ops.nonfinite_scan(dev(z["x"][:1]))[0]) == 0
    assert int(ops.nonfinite_scan(x[:0])[0]) == 0
    rec2 = ops.nonfinite_scan(x[3:, :99].contiguous()).cpu().numpy()      # T = 99: 99 * 41 floats per system, not a multiple of 4
    want = {b - 3 for b in (CERTAIN | EXACT) if b >= 3} - {11 - 3, 13 - 3, 5 - 3}     # their damage sat at t = 99
    assert {int(e >> 1) for e in rec2[4:4 + int(rec2[0])]} == want
    # with no column masked, an infinity in a formerly masked column is no longer certain
    rec3 = ops.nonfinite_scan(x, plan=ops.get_plan(0, 0.5)).cpu().numpy()
    assert {int(e >> 1) for e in rec3[4:4 + int(rec3[0])] if e & 1} == {1, 4, 7, 9, 12, 14, 15, 18}


@pytest.mark.parametrize("si", (0, 12))
@pytest.mark.parametrize("mode", ("fused", "workspace", "generic"))
def test_forward_swag_fast_on_damaged_systems(si, mode, ops, orc, z, swag_states):
    """forward_swag_fast (:878-908) with the reference's own normals: NaN where the reference has NaN (every damaged system, and the
    finite 1e30 one whose pool overflows), the clean system within 1e-5; the in-prologue draw, the workspace form and the generic
    engine agree bit for bit -- and without the scan (assume_finite=True) the kernels alone do NOT give the reference's answer."""
    x = dev(z["x"])
    wa, w2, pd = state(swag_states, si)
    t = tp(z, f"v50_{si}_swagfast_tape")
    z1, z2 = dev(t[0]), dev(t[1].reshape(1, -1))
    eps = dev(np.stack([t[2], t[3]], 1)[None])
    idx = torch.zeros(1, dtype=torch.int32)
    kw = dict(single_launch=True) if mode == "fused" else dict(single_launch=False) if mode == "workspace" else dict(engine="generic")
    out, pre, summ = ops.multiswag(x, wa, w2, pd, idx, z1, z2, eps, debug=True, **kw)
    same_nan_close_elsewhere(out[0].cpu().numpy(), z[f"v50_{si}_swagfast_out"])
    assert torch.isnan(pre[0, 1:]).all() and torch.isfinite(pre[0, 0]).all()
    plain = ops.multiswag(x, wa, w2, pd, idx, z1, z2, eps, **kw)          # no side-effect outputs: certain systems are answered directly
    assert torch.equal(torch.nan_to_num(plain, nan=-1.0), torch.nan_to_num(out, nan=-1.0))
    blind = ops.multiswag(x, wa, w2, pd, idx, z1, z2, eps, assume_finite=True, **kw)
    assert torch.equal(blind[0, 0], out[0, 0])                            # clean systems never depend on the scan
    if mode != "generic":   # (the ahead-of-time generic form multiplies masked columns by zero weights: NaN x 0 = NaN there)
        assert torch.isfinite(blind[0, [4, 5, 6, 7, 8, 17]]).all()        # masked columns are never read: finite, i.e. NOT the reference
    # the summary of a damaged system: NaN in the oracle's places (it follows the reference: tests/test_oracle_nonfinite.py)
    w = orc.swag_draw(swag_states[si]["w_avg"], swag_states[si]["w2_avg"], swag_states[si]["pre_D"], t[0], t[1])
    _, ex = orc.forward(z["x"], w, t[2], t[3], sched=orc.make_schedule(None, pool_parts=4), extras=True)
    assert np.array_equal(np.isnan(summ[0].cpu().numpy()), np.isnan(ex["summary"]))


@pytest.mark.parametrize("si", (0, 12))
@pytest.mark.parametrize("noisy", (0, 1))
@pytest.mark.parametrize("engine", ("auto", "generic"))
def test_varmodel_forward_on_damaged_systems(si, noisy, engine, ops, orc, z):
    x, W = dev(z["x"]), dev(z[f"v50_{si}_swagfast_w"][None])
    t = tp(z, f"v50_{si}_forward_noisy{noisy}_tape")
    e1, e2 = (t[1], t[2]) if noisy else (t[0], t[1])
    kw = dict(eps_in=dev(t[0][None]), eps_sum=dev(t[3][None])) if noisy else {}
    out, pre, summ = ops.forward(x, W, eps=dev(np.stack([e1, e2], 1)[None]), debug=True, engine=engine, **kw)
    same_nan_close_elsewhere(out[0].cpu().numpy(), z[f"v50_{si}_forward_noisy{noisy}_out"])
    # _cur_summary (:512) and latents (:433) as the reference leaves them: the same NaN / inf pattern, the clean system's values
    want = z[f"v50_{si}_forward_noisy{noisy}_summary"]
    got = summ[0].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.abs(got[0] - want[0]).max() <= 2e-5 * max(1.0, float(np.abs(want[0]).max()))
    lat = ops.feature_latents(x, W, eps_in=kw.get("eps_in"))[0].cpu().numpy()
    wl = z[f"v50_{si}_forward_noisy{noisy}_latents"]
    assert np.array_equal(np.isnan(lat[:4]), np.isnan(wl)) and np.array_equal(np.isinf(lat[:4]), np.isinf(wl))
    assert np.array_equal(np.sign(lat[:4][np.isinf(wl)]), np.sign(wl[np.isinf(wl)]))
    fin = np.isfinite(wl)
    assert np.abs(lat[:4][fin] - wl[fin]).max() <= 2e-5 * max(1.0, float(np.abs(wl[fin]).max()))


def test_an_infinity_that_dies_in_the_relu_leaves_finite_outputs(ops, orc, z):
    """The `dead` network (hidden 20, latent 10: the generic engine): +inf on a live column whose feature_nn.0 weights are all negative
    -> finite outputs within 1e-5 of the reference's; -inf there, NaN there, or a non-finite value in a masked column next to it: NaN."""
    plan, arch = dead_plan(ops, orc, z)
    x = dev(z["dead_x"])
    wa, w2, pd = dev(z["dead_w_avg"][None]), dev(z["dead_w2_avg"][None]), dev(z["dead_pre_D"][None])
    t = tp(z, "dead_swagfast_tape")
    idx = torch.zeros(1, dtype=torch.int32)
    out = ops.multiswag(x, wa, w2, pd, idx, dev(t[0]), dev(t[1].reshape(1, -1)), dev(np.stack([t[2], t[3]], 1)[None]), plan=plan)
    same_nan_close_elsewhere(out[0].cpu().numpy(), z["dead_swagfast_out"])
    assert torch.isfinite(out[0, [0, 1, 3, 5, 7]]).all() and torch.isnan(out[0, [2, 4, 6]]).all()
    rec = ops.nonfinite_scan(x, plan=plan).cpu().numpy()
    assert {int(e >> 1): int(e & 1) for e in rec[4:4 + int(rec[0])]} == {1: 0, 2: 0, 3: 0, 4: 1, 5: 0, 6: 1}
    W = dev(z["dead_swagfast_w"][None])
    for noisy in (0, 1):
        t = tp(z, f"dead_forward_noisy{noisy}_tape")
        e1, e2 = (t[1], t[2]) if noisy else (t[0], t[1])
        kw = dict(eps_in=dev(t[0][None]), eps_sum=dev(t[3][None])) if noisy else {}
        out, pre, summ = ops.forward(x, W, eps=dev(np.stack([e1, e2], 1)[None]), plan=plan, debug=True, **kw)
        same_nan_close_elsewhere(out[0].cpu().numpy(), z[f"dead_forward_noisy{noisy}_out"])
        want = z[f"dead_forward_noisy{noisy}_summary"]
        assert np.array_equal(np.isnan(summ[0].cpu().numpy()), np.isnan(want))
        # the exact route against the oracle on the same normals: the finite systems it evaluated (1, 3, 5) within 1e-5
        okw = dict(eps_in=t[0], eps_sum=t[3]) if noisy else {}
        o = orc.forward(z["dead_x"], z["dead_swagfast_w"], e1, e2, arch=arch, **okw)
        same_nan_close_elsewhere(out[0].cpu().numpy(), o)
    # the same with every normal generated in-kernel (Philox): the exact route reads the same streams as the kernels
    for noisy in (False, True):
        out = ops.forward(x, W, plan=plan, philox_seed=77, draw_id0=3, system_id0=1000, noisy=noisy)
        eps = ops.philox_normal(2, 77, 3, 1, B=8, system_id0=1000, width=arch.latent).cpu().numpy()[0]
        okw = {}
        if noisy:
            okw = dict(eps_in=ops.philox_normal(3, 77, 3, 1, B=8, system_id0=1000, width=100).cpu().numpy()[0],
                       eps_sum=ops.philox_normal(4, 77, 3, 1, B=8, system_id0=1000, width=2 * arch.latent).cpu().numpy()[0])
        o = orc.forward(z["dead_x"], z["dead_swagfast_w"], eps[:, 0], eps[:, 1], arch=arch, **okw)
        same_nan_close_elsewhere(out[0].cpu().numpy(), o, rtol=2e-5 if noisy else 1e-5)
        assert torch.isfinite(out[0, [1, 3, 5]]).all()


def test_clean_systems_do_not_notice_and_damaged_ones_are_nan_in_every_driver(ops, z, swag_states):
    """In-kernel noise, chunked draws (torch.chunk semantics), a sharded batch, the statistics tail and the slab drivers: systems
    without damage get the bits of a run on the clean batch; damaged ones NaN (the statistics: the NaN bin)."""
    clean = np.tile(load_golden("inputs.npz")["x_slow"], (3, 1, 1))[:77]
    bad = clean.copy()
    hurt = {5: (17, 9, np.nan), 22: (3, 12, np.inf), 30: (0, 3, np.nan), 31: (99, 38, -np.inf), 64: (50, 1, np.inf), 76: (7, 20, np.nan)}
    for b, (t_, c, v) in hurt.items():
        bad[b, t_, c] = v
    keep = [b for b in range(77) if b not in hurt]
    wa = dev(np.stack([swag_states[i]["w_avg"] for i in (0, 12)]))
    w2 = dev(np.stack([swag_states[i]["w2_avg"] for i in (0, 12)]))
    pd = dev(np.stack([swag_states[i]["pre_D"] for i in (0, 12)]))
    idx = torch.arange(30, dtype=torch.int32) % 2
    xc, xb = dev(clean), dev(bad)
    for kw in (dict(), dict(nchunks=10), dict(nchunks=10, single_launch=True), dict(nchunks=3, engine="generic")):
        want = ops.multiswag(xc, wa, w2, pd, idx, philox_seed=9, assume_finite=True, **kw)
        got = ops.multiswag(xb, wa, w2, pd, idx, philox_seed=9, **kw)
        assert torch.equal(got[:, keep], want[:, keep]), kw
        assert torch.isnan(got[:, sorted(hurt)]).all(), kw
        assert torch.equal(ops.multiswag(xc, wa, w2, pd, idx, philox_seed=9, **kw), want)      # a clean batch: the scan changes nothing
    # a shard of the batch (rows 20..60 of 77 with the chunks of the whole batch): the record indexes the shard's own rows
    want = ops.multiswag(xb, wa, w2, pd, idx, philox_seed=9, nchunks=10)
    got = ops.multiswag(xb[20:60].contiguous(), wa, w2, pd, idx, philox_seed=9, nchunks=10, chunk_B=77, chunk_off=20, system_id0=20)
    assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(want[:, 20:60], nan=-1.0))
    # one record for many calls on the same x
    rec = ops.nonfinite_scan(xb)
    assert torch.equal(torch.nan_to_num(ops.multiswag(xb, wa, w2, pd, idx, philox_seed=9, nchunks=10, nonfinite=rec), nan=-1.0),
                       torch.nan_to_num(want, nan=-1.0))
    with pytest.raises(ValueError):
        ops.multiswag(xb[:5].contiguous(), wa, w2, pd, idx, nonfinite=rec)
    # the statistics tail: fused == stats_draw on the pairs, NaN for the damaged systems
    st = ops.stats_params()
    t_f = ops.multiswag_stats(xb, wa, w2, pd, idx, st=st, philox_seed=9, nchunks=10)
    t_m = ops.stats_draw(want, st=st, philox_seed=9)
    assert torch.equal(torch.nan_to_num(t_f, nan=-1.0), torch.nan_to_num(t_m, nan=-1.0)) and torch.isnan(t_f[:, sorted(hurt)]).all()
    # slab drivers: moments of the clean systems as on the clean batch, NaN rows for the damaged ones; bands likewise (NaN percentiles)
    m_b = ops.multiswag_moments(xb, wa, w2, pd, idx, philox_seed=9, draws_per_launch=8)
    m_c = ops.multiswag_moments(xc, wa, w2, pd, idx, philox_seed=9, draws_per_launch=8, assume_finite=True)
    assert torch.equal(m_b[keep], m_c[keep]) and torch.isnan(m_b[sorted(hurt)]).all()
    sk_b, sk_c = ops.QuantileSketch(77), ops.QuantileSketch(77)
    ops.multiswag_bands(xb, wa, w2, pd, idx, sk_b, philox_seed=9, draws_per_launch=8)
    ops.multiswag_bands(xc, wa, w2, pd, idx, sk_c, philox_seed=9, draws_per_launch=8, assume_finite=True)
    pb, pc = sk_b.percentiles((16.0, 50.0, 84.0)), sk_c.percentiles((16.0, 50.0, 84.0))
    assert torch.equal(pb[keep], pc[keep]) and torch.isnan(pb[sorted(hurt)]).all()
    # the reduced-precision forms keep their own arithmetic for clean systems; damaged ones get the fp32 reference answer
    lp = ops.multiswag(xb, wa, w2, pd, idx, philox_seed=9, precision="bf16x6")
    assert torch.isnan(lp[:, sorted(hurt)]).all() and torch.isfinite(lp[:, keep]).all()


def test_the_module_surface_on_damaged_systems(z, tmp_path):
    """load_swag -> forward_swag_fast / forward / sample under the reference's seeds: the reference's NaNs, by default; the side effects
    _cur_summary and latents with the reference's pattern; assume_finite = True is the opt-out."""
    from bnn_chaos_model_amd import checkpoint
    from bnn_chaos_model_amd import spock_reg_model as srm
    z0 = load_golden("swag_v50_0.npz")
    path = str(tmp_path / "steps=300000_v50_00_output.pkl")
    checkpoint.write_swag_file(path, json.loads(str(z0["hparams_json"])), json.loads(str(z0["swa_params_json"])), torch.tensor(z0["w_avg"]),
                               torch.tensor(z0["w2_avg"]), torch.tensor(z0["pre_D"]))
    m = srm.load_swag(path).cpu().eval()
    assert m.assume_finite is False
    x = torch.tensor(z["x"])
    torch.manual_seed(9100)
    out = m.forward_swag_fast(x, scale=0.5)
    same_nan_close_elsewhere(out.numpy(), z["v50_0_swagfast_out"])
    lat = m.latents                                                 # the weights the fused kernel drew, re-drawn on demand
    wl = z["v50_0_forward_noisy0_latents"]
    assert lat.shape == (20, 100, 20) and np.array_equal(np.isnan(lat[:4].cpu().numpy()), np.isnan(wl))
    for noisy in (False, True):
        torch.manual_seed(9101 + int(noisy))
        out = m(x, noisy_val=noisy)
        same_nan_close_elsewhere(out.numpy(), z[f"v50_0_forward_noisy{int(noisy)}_out"])
        assert np.array_equal(np.isnan(m._cur_summary.cpu().numpy()), np.isnan(z[f"v50_0_forward_noisy{int(noisy)}_summary"]))
    m.load(torch.tensor(z["sample_w"]))
    torch.manual_seed(9300)
    np.random.seed(9300)
    s = m.sample(x, samples=2)
    assert np.array_equal(np.isnan(s), np.isnan(z["sample_out"])) and abs(s[0] - z["sample_out"][0]) <= 2e-5 * abs(z["sample_out"][0])
    m.assume_finite = True
    torch.manual_seed(9100)
    blind = m.forward_swag_fast(x, scale=0.5)
    assert np.abs(blind[0].numpy() - z["v50_0_swagfast_out"][0]).max() <= 1e-5 * 12 and torch.isfinite(blind[4:9]).all()


def test_scan_at_scale_costs_one_pass_over_x(ops, swag_states):
    """2 x 10^5 systems (3.3 GB): the scan lists exactly the planted systems; its time is that of one streaming read of x."""
    B = 200_000
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((B, 100, 41), device="cuda", generator=g)
    planted = torch.randint(0, B, (500,), generator=torch.Generator().manual_seed(4)).unique()
    cols = torch.randint(0, 41, (planted.numel(),), generator=torch.Generator().manual_seed(5))
    ts = torch.randint(0, 100, (planted.numel(),), generator=torch.Generator().manual_seed(6))
    vals = torch.tensor([float("nan"), float("inf"), float("-inf")])[torch.arange(planted.numel()) % 3]
    x[planted.cuda(), ts.cuda(), cols.cuda()] = vals.cuda()
    rec = ops.nonfinite_scan(x)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        rec = ops.nonfinite_scan(x)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 5
    r = rec.cpu().numpy()
    assert sorted(r[4:4 + int(r[0])] >> 1) == sorted(planted.tolist())
    gbs = x.numel() * 4 / ms / 1e6
    print(f"non-finite scan: {ms:.3f} ms for {x.numel() * 4 / 1e9:.2f} GB = {gbs:.0f} GB/s")
    assert gbs > 0
    if os.environ.get("BNN_PERF_ASSERTS") == "1":   # a RATE check is box-dependent (shared / power-capped leases): opt-in, not part of the correctness suite
        assert gbs > 1500.0  # a streaming read (the chip's copy rate is ~6 TB/s); anything far below says the loads are not coalesced
    # and the whole call: J = 20 draws with 0.2 % of the systems damaged -- NaN there, everything else as on clean data
    wa, w2, pd = (dev(swag_states[0][k][None]) for k in ("w_avg", "w2_avg", "pre_D"))
    idx = torch.zeros(20, dtype=torch.int32)
    got = ops.multiswag(x, wa, w2, pd, idx, philox_seed=1)
    keep = torch.ones(B, dtype=torch.bool)
    keep[planted] = False
    assert torch.isfinite(got[:, keep.cuda()]).all()
    nan_rows = torch.isnan(got).any(0).any(-1).cpu()
    certain = torch.isnan(vals) | torch.tensor([(ops.V50_ZERO_MASK >> int(c)) & 1 == 1 for c in cols])
    assert nan_rows[planted[certain]].all() and not nan_rows[keep].any()


def _arch_case(ops, orc, name):
    """plan / oracle arch / weights / inputs of a reference-built network of tests/golden/case_arch_<name>.npz (tests/test_hip_arch.py)."""
    za = load_golden(f"case_arch_{name}.npz")
    hp = json.loads(str(za["hparams_json"]))
    for k, v in list(hp.items()):
        if isinstance(v, str) and v in ("True", "False"):
            hp[k] = v == "True"
    mask = ops.zero_mask_from_flags(hp.get("fix_megno", False), hp.get("fix_megno2", False), hp["include_mmr"], hp["include_nan"], hp.get("include_eplusminus", True))
    kw = dict(n_features=int(za["n_features"]), hidden=hp["hidden"], latent=hp["latent"], depth_in=hp["in"], depth_out=hp["out"])
    plan = ops.get_plan(mask, 0.5, fix_megno=hp.get("fix_megno", False), **kw)
    arch = orc.make_arch(T=100, zero_mask=mask, lowest=0.5, fix_megno=hp.get("fix_megno", False), **kw)
    return plan, arch, za["swagfast_w"], za["x"]


@pytest.mark.parametrize("net", ("v50", "dead", "h48megno", "deriv82", "lin0out8"))
@pytest.mark.parametrize("noisy", (False, True))
def test_random_damage_against_the_oracle(net, noisy, ops, orc, z, swag_states):
    """256 systems, each with 0-3 random damages (NaN / +inf / -inf at a random timestep and column) -- and, for the `dead` network, extra
    +inf on its dead column --, explicit noise: the HIP path against the oracle (which follows the reference, tests/test_oracle_nonfinite.py):
    the same NaN pattern in (mu, std), 1e-5 where finite; systems without damage bit-identical to a run on the clean batch."""
    rng = np.random.default_rng(11 + int(noisy))
    B = 256
    if net in ("v50", "dead"):
        base = np.tile(load_golden("inputs.npz")["x_slow"], (8, 1, 1))[:B].copy()
    else:   # fix_megno (the raw MEGNO column is summarised before it is zeroed), 82 features, a ten-module regress_nn: the generic engine
        plan, arch, w, xa = _arch_case(ops, orc, net)
        base = np.tile(xa, (B // xa.shape[0] + 1, 1, 1))[:B].copy()
    NF = base.shape[2]
    base += 0.01 * rng.standard_normal(base.shape).astype(np.float32)
    x = base.copy()
    hurt = set()
    vals = (np.nan, np.inf, -np.inf)
    for b in range(B):
        for _ in range(int(rng.integers(0, 4)) if b % 3 else 0):
            x[b, int(rng.integers(0, 100)), int(rng.integers(0, NF))] = vals[int(rng.integers(0, 3))]
            hurt.add(b)
    if net not in ("v50", "dead"):
        pass
    elif net == "dead":
        plan, arch = dead_plan(ops, orc, z)
        w = z["dead_swagfast_w"]
        col = int(z["dead_col"])
        for b in range(1, B, 7):      # +inf on the dead column only: dies in the ReLU (unless the system is otherwise damaged)
            x[b, int(rng.integers(0, 100)), col] = np.inf
            hurt.add(b)
    else:
        plan, arch = ops.get_plan(), orc.make_arch(T=100)
        w = z["v50_0_swagfast_w"]
    L, SM = arch.latent, 2 * arch.latent + 2 * int(arch.fix_megno)
    e1, e2 = rng.standard_normal((B, L), dtype=np.float32), rng.standard_normal((B, L), dtype=np.float32)
    kw, okw = {}, {}
    if noisy:
        e_in, e_sum = rng.standard_normal((B, 100, NF), dtype=np.float32), rng.standard_normal((B, SM), dtype=np.float32)
        kw, okw = dict(eps_in=dev(e_in[None]), eps_sum=dev(e_sum[None])), dict(eps_in=e_in, eps_sum=e_sum)
    eps = dev(np.stack([e1, e2], 1)[None])
    W = dev(w[None])
    got = ops.forward(dev(x), W, eps=eps, plan=plan, **kw)[0].cpu().numpy()
    want = orc.forward(x, w, e1, e2, arch=arch, **okw)
    same_nan_close_elsewhere(got, want, rtol=2e-5 if noisy else 1e-5)
    clean = ops.forward(dev(base), W, eps=eps, plan=plan, assume_finite=True, **kw)[0].cpu().numpy()
    keep = np.array([b for b in range(B) if b not in hurt])
    assert np.array_equal(got[keep], clean[keep])
    nan_rows = np.isnan(got).any(1)
    assert not nan_rows[keep].any() and nan_rows.sum() >= len(hurt) // 2
    if net == "dead":   # some of the +inf-on-the-dead-column systems stay finite (those with no other damage): the exact route's work
        only_dead = [b for b in range(1, B, 7) if b % 3 == 0 or np.isfinite(np.delete(x[b], col, axis=1)).all()]
        finite_dead = [b for b in only_dead if np.isfinite(want[b]).all()]
        assert len(finite_dead) >= 3 and np.isfinite(got[finite_dead]).all()
