"""The generic forward engine on the GPU: the network the reference builds from hparams (spock_reg_model.py:301-321, 346-362: any
hidden / latent, depth `in` / `out`, 41 or 82 features, fix_megno), any series length T >= 2 (:416-435), SWAG rank above 32.
Against fixtures the UNMODIFIED reference produced with those shapes (tests/golden/make_golden_arch.py) at 1e-5 relative with zero
exceedances, and against the oracle on the engine's accumulation schedule (natural order, four strided pool partitions) bit for bit.
Needs an MI355X."""
import json

import numpy as np
import pytest
import torch

from conftest import close_report, load_golden

pytestmark = pytest.mark.gpu

CASES = ("h64l16", "h20l10", "h33l7", "deep22", "deep30", "lin00", "deriv82", "k40", "h48megno", "h128l32", "allcols", "lin0out8")
TLENS = (2, 3, 5, 6, 7, 99)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def tp(z, pfx):
    return [z[f"{pfx}_{i:03d}"] for i in range(int(z[pfx + "_n"]))]


def hparams_of(z):
    hp = json.loads(str(z["hparams_json"]))
    for k, v in list(hp.items()):
        if isinstance(v, str) and v in ("True", "False"):
            hp[k] = v == "True"
    return hp


@pytest.fixture(scope="module")
def ops():
    from bnn_chaos_model_amd import ops as o
    return o


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def orc_mod():
    from oracle import oracle
    return oracle


def plan_and_arch(ops, orc, z, T=100):
    hp = hparams_of(z)
    mask = ops.zero_mask_from_flags(hp.get("fix_megno", False), hp.get("fix_megno2", False), hp["include_mmr"], hp["include_nan"],
                                    hp.get("include_eplusminus", True))
    lowest = 0.1 if hp.get("lower_std", False) else 0.5
    kw = dict(n_features=int(z["n_features"]), hidden=hp["hidden"], latent=hp["latent"], depth_in=hp["in"], depth_out=hp["out"])
    plan = ops.get_plan(mask, lowest, fix_megno=hp.get("fix_megno", False), **kw)
    arch = orc.make_arch(T=T, zero_mask=mask, lowest=lowest, fix_megno=hp.get("fix_megno", False), **kw)
    return plan, arch


@pytest.mark.parametrize("name", CASES)
def test_forward_vs_reference_and_oracle(name, ops, orc):
    z = load_golden(f"case_arch_{name}.npz")
    plan, arch = plan_and_arch(ops, orc, z)
    assert plan.d == z["w_avg"].size == orc.param_count(arch)
    assert not plan.v50net or name == "k40"
    x, W = dev(z["x"]), dev(z["swagfast_w"][None])
    B = x.shape[0]
    sched = orc.make_schedule(None, pool_parts=4)
    for noisy in (0, 1):
        t = tp(z, f"forward_noisy{noisy}_tape")
        e1, e2 = (t[1], t[2]) if noisy else (t[0], t[1])
        eps = dev(np.stack([e1, e2], 1)[None])
        kw = dict(eps_in=dev(t[0][None]), eps_sum=dev(t[3][None])) if noisy else {}
        out, pre, summ = ops.forward(x, W, eps=eps, plan=plan, debug=True, engine="generic", **kw)
        nbad, mx = close_report(out[0].cpu().numpy(), z[f"forward_noisy{noisy}_out"])          # the reference itself, 1e-5 relative
        assert nbad == 0, (name, noisy, nbad, mx)
        okw = dict(eps_in=t[0], eps_sum=t[3]) if noisy else {}
        o, ex = orc.forward(z["x"], z["swagfast_w"], e1, e2, arch=arch, sched=sched, extras=True, **okw)
        assert summ.shape == (1, B, plan.summary_width)
        if noisy:   # expf of the noise scales differs by an ulp between libm and the device
            nbad, mx = close_report(summ[0].cpu().numpy(), ex["summary"], rtol=2e-6, atol=2e-6 * max(1.0, float(np.abs(ex["summary"]).max())))
            assert nbad == 0, (name, nbad, mx)
        else:   # same IEEE operations in the same order: summary and pre-clamp outputs bit for bit
            assert np.array_equal(summ[0].cpu().numpy(), ex["summary"]), name
            assert np.array_equal(pre[0].cpu().numpy(), ex["pre_clamp"]), name
        assert np.abs(out[0].cpu().numpy() - o).max() <= 2e-6      # tanhf: libm vs device
        # the self.latents side effect (:417, :433): feature_nn per time step, bit for bit (and the forward beside it is unchanged)
        lat = ops.feature_latents(x, W, eps_in=dev(t[0][None]) if noisy else None, plan=plan)
        assert lat.shape == (1, B, x.shape[1], plan.latent)
        if noisy:   # (expf of the input-noise scale: an ulp between libm and the device)
            nbad, mx = close_report(lat[0].cpu().numpy(), ex["latents"], rtol=2e-6, atol=2e-6 * max(1.0, float(np.abs(ex["latents"]).max())))
            assert nbad == 0, (name, nbad, mx)
        else:
            assert np.array_equal(lat[0].cpu().numpy(), ex["latents"]), name


@pytest.mark.parametrize("name", CASES)
def test_multiswag_draw_regress_and_philox(name, ops, orc):
    """forward_swag_fast (:878-908) through the draw-once form; the draw equals the oracle's bit for bit (any d, K = 40 and 5 and 6
    included); in-kernel Philox == explicit tensors; predict_instability on the explicit summary == the kernel's tail."""
    z = load_golden(f"case_arch_{name}.npz")
    plan, arch = plan_and_arch(ops, orc, z)
    t = tp(z, "swagfast_tape")
    wa, w2, pd = dev(z["w_avg"][None]), dev(z["w2_avg"][None]), dev(z["pre_D"][None])
    idx = torch.zeros(1, dtype=torch.int32)
    z1, z2 = dev(t[0]), dev(t[1].reshape(1, -1))
    W = ops.swag_draw(wa, w2, pd, idx, z1, z2, scale=0.5, plan=plan)
    assert np.array_equal(W[0].cpu().numpy(), orc.swag_draw(z["w_avg"], z["w2_avg"], z["pre_D"], t[0], t[1]))
    assert np.abs(W[0].cpu().numpy() - z["swagfast_w"]).max() <= 2e-6
    eps = dev(np.stack([t[2], t[3]], 1)[None])
    x = dev(z["x"])
    out = ops.multiswag(x, wa, w2, pd, idx, z1, z2, eps, plan=plan, engine="generic")
    nbad, mx = close_report(out[0].cpu().numpy(), z["swagfast_out"])
    assert nbad == 0, (name, nbad, mx)
    if not ops.fused_draw_available(plan, x.shape[1], pd.shape[2]):
        with pytest.raises(NotImplementedError):
            ops.multiswag(x, wa, w2, pd, idx, z1, z2, eps, plan=plan, single_launch=True)
    # Philox noise in the kernel == the same numbers as explicit tensors (quiet and noisy), at global offsets
    B, T, NF = x.shape
    R, seed = 3, 1234
    Wr = W.expand(R, -1).contiguous()
    a = ops.forward(x, Wr, philox_seed=seed, draw_id0=6, system_id0=777, noisy=True, plan=plan, engine="generic")
    e = ops.philox_normal(2, seed, 6, R, width=plan.latent, B=B, system_id0=777)
    e_in = ops.philox_normal(3, seed, 6, R, width=T, B=B, system_id0=777, n_features=NF)
    e_sum = ops.philox_normal(4, seed, 6, R, width=plan.summary_width, B=B, system_id0=777)
    b = ops.forward(x, Wr, eps=e, eps_in=e_in, eps_sum=e_sum, plan=plan, engine="generic")
    assert torch.equal(a, b), name
    q, pre_q, summ_q = ops.forward(x, Wr, philox_seed=seed, draw_id0=6, system_id0=777, plan=plan, debug=True, engine="generic")
    q2 = ops.forward(x, Wr, eps=e, plan=plan, engine="generic")
    assert torch.equal(q, q2)
    out_r, pre_r = ops.regress(summ_q, Wr, plan=plan, debug=True)
    if not plan.v50net:   # the pretrained network's regress kernel follows ITS forward kernel's (permuted) order
        assert torch.equal(pre_r, pre_q) and torch.equal(out_r, q), name
    # the statistics tail fused in the generic kernel == the stand-alone epilogue on its (mu, std)
    tq = ops.multiswag_stats(x, wa, w2, pd, idx.expand(R).contiguous(), philox_seed=seed, draw_id0=6, system_id0=777, plan=plan)
    ms = ops.multiswag(x, wa, w2, pd, idx.expand(R).contiguous(), philox_seed=seed, draw_id0=6, system_id0=777, plan=plan, engine="generic")
    assert torch.equal(tq, ops.stats_draw(ms, philox_seed=seed, row_id0=6, system_id0=777)) or plan.v50net


@pytest.mark.parametrize("T", TLENS)
def test_series_lengths(T, ops, orc, swag_states):
    """x[:, :T] through the pretrained member v50_0: the pretrained network's kernels take T % 4 == 0, T >= 8; every other length runs
    on the generic engine (per-lane counts, unequal-count merges)."""
    z = load_golden("case_arch_tlen.npz")
    st = swag_states[0]
    plan = ops.get_plan()
    x = dev(z["x"][:, :T])
    t = tp(z, f"T{T}_tape")
    wa, w2, pd = dev(st["w_avg"][None]), dev(st["w2_avg"][None]), dev(st["pre_D"][None])
    idx = torch.zeros(1, dtype=torch.int32)
    eps = dev(np.stack([t[2], t[3]], 1)[None])
    out, pre, summ = ops.multiswag(x, wa, w2, pd, idx, dev(t[0]), dev(t[1].reshape(1, -1)), eps, plan=plan, debug=True)
    nbad, mx = close_report(out[0].cpu().numpy(), z[f"T{T}_out"], rtol=2e-5 if T == 2 else 1e-5)   # T = 2: tests/test_oracle_arch.py
    assert nbad == 0, (T, nbad, mx)
    arch = orc.make_arch(T=T)
    w = orc.swag_draw(st["w_avg"], st["w2_avg"], st["pre_D"], t[0], t[1])
    o, ex = orc.forward(z["x"][:, :T], w, t[2], t[3], arch=arch, sched=orc.make_schedule(None, pool_parts=4), extras=True)
    assert np.array_equal(summ[0].cpu().numpy(), ex["summary"]), T
    assert np.array_equal(pre[0].cpu().numpy(), ex["pre_clamp"]), T
    tn = tp(z, f"T{T}_noisy_tape")
    W = dev(z[f"T{T}_w"][None])
    on = ops.forward(x, W, eps=dev(np.stack([tn[1], tn[2]], 1)[None]), eps_in=dev(tn[0][None]), eps_sum=dev(tn[3][None]), plan=plan)
    nbad, mx = close_report(on[0].cpu().numpy(), z[f"T{T}_noisy_out"], rtol=2e-5 if T == 2 else 1e-5)
    assert nbad == 0, (T, nbad, mx)


def test_generic_engine_on_the_pretrained_network_equals_its_kernels(ops, swag_states, inputs):
    """Same decomposition, same accumulation order in feature_nn and the pool: on the v50 network at T = 100 the generic engine's
    summary is bit-identical to the register-resident kernels'; regress_nn's orders differ (natural vs k-step major), so the outputs
    agree to rounding."""
    st = swag_states[12]
    x = dev(np.tile(inputs["slow"], (5, 1, 1))[:150])
    rng = np.random.default_rng(3)
    W = dev(st["w_avg"][None] + 0.01 * rng.standard_normal((4, st["w_avg"].size)).astype(np.float32))
    for mask in (ops.V50_ZERO_MASK, 0, 1 << 7):
        plan = ops.get_plan(mask)
        for noisy in (False, True):
            a = ops.forward(x, W, philox_seed=5, draw_id0=8, system_id0=99, plan=plan, debug=True, noisy=noisy)
            b = ops.forward(x, W, philox_seed=5, draw_id0=8, system_id0=99, plan=plan, debug=True, noisy=noisy, engine="generic")
            assert torch.equal(a[2], b[2]), (mask, noisy)
            nbad, mx = close_report(b[0].cpu().numpy(), a[0].cpu().numpy(), rtol=5e-6)   # regress_nn: two summation orders of 40-term sums
            assert nbad == 0, (mask, noisy, nbad, mx)


def test_generic_engine_at_scale_invariances(ops, orc):
    """20 000 systems x 6 draws of the (64, 16) network with in-kernel Philox: spot checks against the oracle fed the kernel's own
    normals; results do not depend on the block size, on system sharding (global ids) or on draw slabs."""
    z = load_golden("case_arch_h64l16.npz")
    plan, arch = plan_and_arch(ops, orc, z)
    rng = np.random.default_rng(11)
    B, J, seed = 20000, 6, 77
    xb = np.tile(z["x"], (B // 16 + 1, 1, 1))[:B] * rng.uniform(0.8, 1.2, (B, 1, 1)).astype(np.float32)
    x = dev(xb)
    W = dev(z["swagfast_w"][None] + 0.02 * rng.standard_normal((J, plan.d)).astype(np.float32))
    full = ops.forward(x, W, philox_seed=seed, draw_id0=12, system_id0=5000, plan=plan)
    for spb in (64, 256):
        assert torch.equal(full, ops.forward(x, W, philox_seed=seed, draw_id0=12, system_id0=5000, plan=plan, systems_per_block=spb))
    lo = ops.forward(x[:7001], W, philox_seed=seed, draw_id0=12, system_id0=5000, plan=plan)
    hi = ops.forward(x[7001:], W, philox_seed=seed, draw_id0=12, system_id0=5000 + 7001, plan=plan)
    assert torch.equal(full, torch.cat([lo, hi], 1))
    s2 = ops.forward(x, W[2:4], philox_seed=seed, draw_id0=14, system_id0=5000, plan=plan)
    assert torch.equal(full[2:4], s2)
    ch = ops.forward(x, W, philox_seed=seed, draw_id0=12, system_id0=5000, plan=plan, nchunks=3)   # torch.chunk semantics: 2 output rows
    assert ch.shape == (2, B, 2)
    eps = ops.philox_normal(2, seed, 12, J, width=plan.latent, B=B, system_id0=5000).cpu().numpy()
    Wn, fulln = W.cpu().numpy(), full.cpu().numpy()
    sched = orc.make_schedule(None, pool_parts=4)
    for (j, b) in [(0, 0), (5, B - 1), (2, 7000), (3, 7001), (1, 12345), (4, 63), (4, 64)]:
        o = orc.forward(xb[b:b + 1], Wn[j], eps[j, b:b + 1, 0], eps[j, b:b + 1, 1], arch=arch, sched=sched)
        assert np.abs(fulln[j, b] - o[0]).max() <= 2e-6, (j, b)


def test_surface_accepts_checkpoints_of_other_shapes(tmp_path, ops):
    """load_swag on checkpoints whose hparams describe other networks -> the module API replays the reference run seed for seed
    (forward_swag_fast :878-908, forward :486-528); state_dict keys / shapes are the reference's."""
    from bnn_chaos_model_amd import checkpoint, spock_reg_model as srm
    for name in ("h64l16", "deep22", "lin00", "deriv82", "h33l7", "k40", "h48megno"):
        z = load_golden(f"case_arch_{name}.npz")
        hp = hparams_of(z)
        p = tmp_path / f"{name}_output.pkl"
        checkpoint.write_swag_file(str(p), hp, json.loads(str(z["swa_params_json"])), torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]),
                                   torch.tensor(z["pre_D"]))
        m = srm.load_swag(str(p)).eval()
        sd = m.state_dict()
        assert list(sd.keys()) == [str(k) for k in z["state_keys"]]
        assert [list(v.shape) for v in sd.values()] == json.loads(str(z["state_shapes"]))
        x = torch.tensor(z["x"])
        torch.manual_seed(hp["seed"] + 2)
        out = m.forward_swag_fast(x, scale=0.5)
        nbad, mx = close_report(out.numpy(), z["swagfast_out"])
        assert nbad == 0, (name, nbad, mx)
        assert np.abs(m.flatten().numpy() - z["swagfast_w"]).max() <= 2e-6        # the sampled weights stay loaded (:838)
        _, arch = plan_and_arch(ops, orc_mod(), z)
        zero = np.zeros((x.shape[0], hp["latent"]), np.float32)
        sched = orc_mod().make_schedule(None, pool_parts=4)
        _, ex = orc_mod().forward(z["x"], m.flatten().numpy(), zero, zero, arch=arch, sched=sched, extras=True)
        assert np.array_equal(m.latents.cpu().numpy(), ex["latents"]), name        # ... and so do the latents of that call (:433)
        for noisy in (False, True):
            torch.manual_seed(hp["seed"] + 3 + int(noisy))
            o = m(x, noisy_val=noisy)
            nbad, mx = close_report(o.numpy(), z[f"forward_noisy{int(noisy)}_out"])
            assert nbad == 0, (name, noisy, nbad, mx)
            # forward()'s side effect self._cur_summary (:512: the summary before the summary noise), produced on demand
            ref = z[f"forward_noisy{int(noisy)}_summary"]
            cs = m._cur_summary
            assert cs.shape == ref.shape
            nbad, mx = close_report(cs.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
            assert nbad == 0, (name, noisy, nbad, mx)
            # compute_summary_stats' side effect self.latents (:433), on demand as well: feature_nn of THIS call (same input noise)
            t = tp(z, f"forward_noisy{int(noisy)}_tape")
            _, ex = orc_mod().forward(z["x"], m.flatten().numpy(), zero, zero, eps_in=t[0] if noisy else None, eps_sum=None, arch=arch,
                                      sched=sched, extras=True)
            nbad, mx = close_report(m.latents.cpu().numpy(), ex["latents"], rtol=2e-6, atol=2e-6 * max(1.0, float(np.abs(ex["latents"]).max())))
            assert nbad == 0 and (noisy or np.array_equal(m.latents.cpu().numpy(), ex["latents"])), (name, noisy, nbad, mx)
        np.random.seed(0); torch.manual_seed(0)
        s = m.sample(x, samples=3)
        assert s.shape == (x.shape[0],) and np.isfinite(s).all()
        with pytest.raises(NotImplementedError):
            m(x[:, :, :-1])


def test_latents_on_the_pretrained_network(ops, orc, swag_states, inputs):
    """feature_nn alone on the pretrained network (T = 100 and a ragged T) == the oracle's per-step latents bit for bit; in-kernel
    Philox input noise == the explicit-tensor form."""
    wa, w2, pd = (dev(swag_states[0][k][None]) for k in ("w_avg", "w2_avg", "pre_D"))
    plan = ops.get_plan()
    W = ops.swag_draw(wa, w2, pd, torch.zeros(2, dtype=torch.int32), philox_seed=5, plan=plan)
    for T in (100, 37):
        xh = np.ascontiguousarray(inputs["slow"][:, :T])
        x = dev(xh)
        B = x.shape[0]
        lat = ops.feature_latents(x, W, plan=plan)
        zero = np.zeros((B, 20), np.float32)
        for j in range(2):
            _, ex = orc.forward(xh, W[j].cpu().numpy(), zero, zero, arch=orc.make_arch(T=T), sched=orc.make_schedule(None, pool_parts=4), extras=True)
            assert np.array_equal(lat[j].cpu().numpy(), ex["latents"])
        e_in = ops.philox_normal(3, 77, 4, 2, width=T, B=B, system_id0=9, n_features=41)
        a = ops.feature_latents(x, W, noisy=True, philox_seed=77, draw_id0=4, system_id0=9, plan=plan)
        b = ops.feature_latents(x, W, eps_in=e_in, plan=plan)
        assert torch.equal(a, b) and not torch.equal(a, lat)


def test_limits_are_errors_not_wrong_answers(ops):
    from bnn_chaos_model_amd import _native as N
    for kw in (dict(hidden=129), dict(latent=65), dict(n_features=40), dict(depth_in=9, depth_out=9)):
        with pytest.raises(N.NativeError) as ei:
            ops.get_plan(**kw)
        assert ei.value.code == N.ERR_UNSUPPORTED
    plan = ops.get_plan(hidden=64, latent=16)
    x = torch.zeros(4, 1, 41, device="cuda")
    with pytest.raises(N.NativeError):
        ops.forward(x, torch.zeros(1, plan.d, device="cuda"), plan=plan)        # T = 1: torch.std is NaN
    with pytest.raises(Exception):
        ops.forward(torch.zeros(4, 8, 41, device="cuda"), torch.zeros(1, plan.d, device="cuda"), plan=plan, precision="bf16")


def test_random_architecture_sweep_against_the_oracle(ops, orc):
    """Seeded random networks (widths 1..128, depths 0..3, 41 / 82 features, fix_megno, random column masks, T in [2, 45]) with random
    weights: the generic engine's summary and pre-clamp outputs equal the oracle's on its schedule bit for bit, quiet; the noisy forward
    agrees to the rounding of expf.  Covers every register bucket, the eight-, four-, two- and one-wave LDS budgets and the staged
    regress_nn path."""
    rng = np.random.default_rng(20260410)
    seen = set()
    for trial in range(14):
        F = 82 if trial % 5 == 4 else 41
        H = int(rng.choice([1, 3, 8, 17, 40, 48, 49, 64, 77, 96, 100, 128]))
        L = int(rng.choice([1, 2, 5, 16, 20, 31, 48, 63]))
        din, dout = int(rng.integers(0, 4)), int(rng.integers(0, 3))
        megno = bool(rng.integers(0, 2))
        mask = int(rng.integers(0, 1 << 41)) | ((1 << 7) if megno else 0)
        if trial == 0:
            H, L, din, dout, megno, mask = 128, 48, 1, 2, False, 0     # the LDS budget leaves one or two waves; regress_nn staged from L2
        T = int(rng.integers(2, 46))
        try:
            plan = ops.get_plan(mask, 0.5, fix_megno=megno, n_features=F, hidden=H, latent=L, depth_in=din, depth_out=dout)
        except Exception as e:      # outside the LDS budget: an error, never a wrong answer
            assert "LDS" in str(e) or "UNSUPPORTED" in str(e).upper() or "fit" in str(e), e
            continue
        arch = orc.make_arch(T=T, zero_mask=mask, n_features=F, hidden=H, latent=L, fix_megno=megno, depth_in=din, depth_out=dout)
        assert plan.d == orc.param_count(arch)
        B = 21
        x = (rng.standard_normal((B, 1, F)) + 0.2 * rng.standard_normal((B, T, F))).astype(np.float32)
        w = (rng.standard_normal(plan.d) * (0.6 / np.sqrt(max(H, 8)))).astype(np.float32)
        eps = rng.standard_normal((1, B, 2, L)).astype(np.float32)
        out, pre, summ = ops.forward(dev(x), dev(w[None]), eps=dev(eps), plan=plan, debug=True, engine="generic")
        o, ex = orc.forward(x, w, eps[0, :, 0], eps[0, :, 1], arch=arch, sched=orc.make_schedule(None, pool_parts=4), extras=True)
        assert np.array_equal(summ[0].cpu().numpy(), ex["summary"]), (trial, F, H, L, din, dout, megno, T)
        assert np.array_equal(pre[0].cpu().numpy(), ex["pre_clamp"]), (trial, F, H, L, din, dout, megno, T)
        assert np.abs(out[0].cpu().numpy() - o).max() <= 2e-6
        e_in = rng.standard_normal((1, B, T, F)).astype(np.float32)
        e_sum = rng.standard_normal((1, B, plan.summary_width)).astype(np.float32)
        on = ops.forward(dev(x), dev(w[None]), eps=dev(eps), eps_in=dev(e_in), eps_sum=dev(e_sum), plan=plan, engine="generic")[0].cpu().numpy()
        wn = orc.forward(x, w, eps[0, :, 0], eps[0, :, 1], eps_in=e_in[0], eps_sum=e_sum[0], arch=arch, sched=orc.make_schedule(None, pool_parts=4))
        nbad, mx = close_report(on, wn, rtol=5e-6, atol=5e-6)
        assert nbad == 0, (trial, nbad, mx)
        seen.add((F, H, L, din, dout))
    assert len(seen) >= 10
