"""Run-time specialised forms of the generic forward engine (specialize.py): the same kernel source compiled
for ONE network with every shape a constant.  Same accumulation order, same arithmetic as the ahead-of-time form -> BIT-IDENTICAL
outputs (mu, std, pre-clamp, summary, latents-free) on every network of the fixture set, quiet and noisy, at ragged T, with the
statistics tail, and through the surface.  Needs an MI355X and hipcc (the ROCm image has it)."""
import json

import numpy as np
import pytest
import torch

from conftest import close_report, load_golden

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    from bnn_chaos_model_amd import ops as o
    return o


def _plan(ops, hidden, latent, din, dout, nf=41, megno=False, mask=None):
    mask = ops.V50_ZERO_MASK if mask is None else mask
    if megno:
        mask |= 1 << 7
    return ops.get_plan(mask, 0.5, fix_megno=megno, n_features=nf, hidden=hidden, latent=latent, depth_in=din, depth_out=dout)


NETS = [(40, 20, 1, 1, 41, False), (64, 16, 1, 1, 41, False), (33, 7, 1, 1, 41, False), (40, 20, 2, 2, 41, False), (30, 12, 0, 0, 41, False),
        (40, 20, 1, 1, 82, False), (48, 24, 1, 1, 41, True), (72, 20, 1, 1, 41, False)]


FIXTURE_NETS = ("h64l16", "deep22", "deep30", "deriv82", "h48megno", "allcols", "lin00", "h33l7", "h20l10")


def fixture_archs():
    """(hidden, latent, in, out, features, fix_megno, mask) of the reference-generated fixture networks used below."""
    from bnn_chaos_model_amd import ops as o
    out = []
    for name in FIXTURE_NETS:
        z = load_golden(f"case_arch_{name}.npz")
        hp = json.loads(str(z["hparams_json"]))
        for k, v in list(hp.items()):
            if isinstance(v, str) and v in ("True", "False"):
                hp[k] = v == "True"
        mask = o.zero_mask_from_flags(hp.get("fix_megno", False), hp.get("fix_megno2", False), hp["include_mmr"], hp["include_nan"],
                                      hp.get("include_eplusminus", True))
        out.append((hp["hidden"], hp["latent"], hp["in"], hp["out"], int(z["n_features"]), hp.get("fix_megno", False), mask))
    return out


@pytest.fixture(scope="module", autouse=True)
def warm_cache():
    """Compile every form this module attaches, side by side (hipcc subprocesses; nothing to do when the in-tree cache travelled)."""
    from bnn_chaos_model_amd import specialize as S
    S.prewarm(NETS, noisy=(False, True), w8=(None,))
    S.prewarm([(56, 14, 1, 1, 41, False)], noisy=(False,), w8=(False, True))
    S.prewarm([(64, 16, 1, 1, 41, False, 0)], noisy=(False, True), w8=(None,))
    S.prewarm(fixture_archs(), noisy=(False, True), w8=(None,))
    S.prewarm([(24, 6, 1, 1, 41, False)], noisy=(False,), w8=(None,))


@pytest.mark.parametrize("net", NETS, ids=lambda n: "h%dl%d_%d%d_f%d%s" % (n[0], n[1], n[2], n[3], n[4], "_megno" if n[5] else ""))
def test_specialised_form_is_bit_identical_to_the_generic_engine(net, ops):
    H, L, din, dout, NF, megno = net
    plan = _plan(ops, H, L, din, dout, NF, megno)
    ops.specialize(plan)
    assert plan.spec_attached(False) and plan.spec_attached(True)
    g = torch.Generator(device="cuda").manual_seed(H * 1000 + L)
    R = 3
    W = torch.randn(R, plan.d, generator=g, device="cuda") * 0.25
    for T, B in ((100, 70), (37, 33), (2, 5)):
        x = torch.randn(B, T, NF, generator=g, device="cuda")
        for noisy in (False, True):
            kw = dict(philox_seed=11, draw_id0=3, system_id0=17, plan=plan, noisy=noisy, debug=True)
            a = ops.forward(x, W, engine="generic", **kw)
            b = ops.forward(x, W, engine="spec", **kw)
            c = ops.forward(x, W, **kw)                       # "auto" takes the specialised form once attached ...
            for u, v, w in zip(a, b, c):
                assert torch.equal(u, v) and (torch.equal(u, w) or (plan.v50net and T % 4 == 0 and T >= 8)), (net, T, noisy)   # (... except where the pretrained network's own kernels run)
            assert torch.isfinite(a[0]).all()
    # chunked draws + the fused statistics tail
    x = torch.randn(130, 100, NF, generator=g, device="cuda")
    W6 = torch.randn(6, plan.d, generator=g, device="cuda") * 0.25
    a = ops.forward(x, W6, nchunks=3, philox_seed=5, plan=plan, engine="generic")
    b = ops.forward(x, W6, nchunks=3, philox_seed=5, plan=plan, engine="spec")
    assert torch.equal(a, b)
    # the statistics tail fused into the specialised kernel (the default route once attached) == the stand-alone epilogue on the
    # ahead-of-time form's (mu, std); the slab driver's moments likewise
    wa = W6[:2].contiguous()
    w2 = wa ** 2 + 1e-4
    pd = wa[:, :, None] + 0.01 * torch.randn(2, plan.d, 5, generator=g, device="cuda")
    idx = torch.tensor([0, 1, 1, 0, 1, 0], dtype=torch.int32)
    tq = ops.multiswag_stats(x, wa, w2, pd, idx, nchunks=3, philox_seed=3, draw_id0=6, system_id0=77, plan=plan)
    ms = ops.multiswag(x, wa, w2, pd, idx, nchunks=3, philox_seed=3, draw_id0=6, system_id0=77, plan=plan)
    assert torch.equal(tq, ops.stats_draw(ms, philox_seed=3, row_id0=2, system_id0=77))
    if not plan.v50net:   # (the pretrained network at T = 100 runs its own kernels on the default route: another regress_nn order)
        assert torch.equal(ms, ops.multiswag(x, wa, w2, pd, idx, nchunks=3, philox_seed=3, draw_id0=6, system_id0=77, plan=plan, engine="generic"))


def _fixture_plan(ops, z):
    hp = json.loads(str(z["hparams_json"]))
    for k, v in list(hp.items()):
        if isinstance(v, str) and v in ("True", "False"):
            hp[k] = v == "True"
    mask = ops.zero_mask_from_flags(hp.get("fix_megno", False), hp.get("fix_megno2", False), hp["include_mmr"], hp["include_nan"],
                                    hp.get("include_eplusminus", True))
    plan = ops.get_plan(mask, 0.1 if hp.get("lower_std", False) else 0.5, fix_megno=hp.get("fix_megno", False), n_features=int(z["n_features"]),
                        hidden=hp["hidden"], latent=hp["latent"], depth_in=hp["in"], depth_out=hp["out"])
    return plan, hp


@pytest.mark.parametrize("name", FIXTURE_NETS)
def test_specialised_forms_against_the_reference_fixtures(name, ops):
    """The specialised forms -- each fixture's own column mask compiled in, masked columns dropped from layer 0 -- against what the
    UNMODIFIED reference produced for that network (tests/golden/make_golden_arch.py): 1e-5 relative, zero exceedances, quiet and noisy,
    forward and forward_swag_fast; and bit-identical to the ahead-of-time form on the same tapes."""
    z = load_golden(f"case_arch_{name}.npz")
    plan, hp = _fixture_plan(ops, z)
    ops.specialize(plan)
    tp = lambda pfx: [z[f"{pfx}_{i:03d}"] for i in range(int(z[pfx + "_n"]))]
    x, W = dev(z["x"]), dev(z["swagfast_w"][None])
    for noisy in (0, 1):
        t = tp(f"forward_noisy{noisy}_tape")
        e1, e2 = (t[1], t[2]) if noisy else (t[0], t[1])
        eps = dev(np.stack([e1, e2], 1)[None])
        kw = dict(eps_in=dev(t[0][None]), eps_sum=dev(t[3][None])) if noisy else {}
        a = ops.forward(x, W, eps=eps, plan=plan, debug=True, engine="spec", **kw)
        b = ops.forward(x, W, eps=eps, plan=plan, debug=True, engine="generic", **kw)
        nbad, mx = close_report(a[0][0].cpu().numpy(), z[f"forward_noisy{noisy}_out"])
        assert nbad == 0, (name, noisy, nbad, mx)
        for u, v in zip(a, b):
            assert torch.equal(u, v), (name, noisy)
    t = tp("swagfast_tape")
    wa, w2, pd = dev(z["w_avg"][None]), dev(z["w2_avg"][None]), dev(z["pre_D"][None])
    out = ops.multiswag(x, wa, w2, pd, torch.zeros(1, dtype=torch.int32), dev(t[0]), dev(t[1].reshape(1, -1)), dev(np.stack([t[2], t[3]], 1)[None]),
                        plan=plan, engine="spec")
    nbad, mx = close_report(out[0].cpu().numpy(), z["swagfast_out"])
    assert nbad == 0, (name, nbad, mx)


def test_both_wave_forms_and_errors(ops):
    plan = _plan(ops, 56, 14, 1, 1)
    g = torch.Generator(device="cuda").manual_seed(1)
    W = torch.randn(2, plan.d, generator=g, device="cuda") * 0.25
    x = torch.randn(50, 100, 41, generator=g, device="cuda")
    from bnn_chaos_model_amd import _native as N
    with pytest.raises(N.NativeError):
        ops.forward(x, W, plan=plan, engine="spec")           # nothing attached yet: an error, not a silent fallback
    ref = ops.forward(x, W, philox_seed=2, plan=plan, engine="generic")
    for w8 in (False, True):
        ops.specialize(plan, noisy=(False,), w8=w8)
        assert torch.equal(ops.forward(x, W, philox_seed=2, plan=plan, engine="spec"), ref), w8
    with pytest.raises(N.NativeError):
        ops.forward(x, W, plan=plan, engine="spec", noisy=True)   # only the quiet form was attached
    big = _plan(ops, 128, 32, 1, 1)
    with pytest.raises(N.NativeError):
        ops.specialize(big, noisy=(False,), w8=True)              # eight waves' pool state does not fit next to that image


def test_surface_specialize(tmp_path, ops, monkeypatch):
    """load_swag(...).specialize(): forward_swag_fast / forward replay the reference's fixture through the specialised form."""
    from bnn_chaos_model_amd import checkpoint, spock_reg_model as srm
    z = load_golden("case_arch_h64l16.npz")
    hp = json.loads(str(z["hparams_json"]))
    for k, v in list(hp.items()):
        if isinstance(v, str) and v in ("True", "False"):
            hp[k] = v == "True"
    p = tmp_path / "h64_output.pkl"
    checkpoint.write_swag_file(str(p), hp, json.loads(str(z["swa_params_json"])), torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"]))
    m = srm.load_swag(str(p)).eval().specialize()
    x = torch.tensor(z["x"])
    torch.manual_seed(hp["seed"] + 2)
    out = m.forward_swag_fast(x, scale=0.5)
    nbad, mx = close_report(out.numpy(), z["swagfast_out"])
    assert nbad == 0, (nbad, mx)
    for noisy in (False, True):
        torch.manual_seed(hp["seed"] + 3 + int(noisy))
        o = m(x, noisy_val=noisy)
        nbad, mx = close_report(o.numpy(), z[f"forward_noisy{int(noisy)}_out"])
        assert nbad == 0, (noisy, nbad, mx)
    assert m._plan().spec_attached(False) and m._plan().spec_attached(True)
    other = m._plan(zero_mask=0)                       # every plan the model hands out afterwards is specialised the same way
    assert other is not m._plan() and other.spec_attached(False) and other.spec_attached(True)
    monkeypatch.setenv("BNN_AUTO_SPECIALIZE", "1")     # the same for every model of the process, without touching the script
    m2 = srm.load_swag(str(p)).eval()
    assert m2._spec is not None and m2._plan(zero_mask=1 << 3).spec_attached(False)


def test_pretrained_network_ragged_T_runs_on_the_embedded_forms(ops, swag_states, inputs):
    """The pretrained network's two specialised forms are compiled INTO the library (bnn_fwd_v50spec.hip): at ragged series lengths the
    default route takes them with nothing attached and no compiler at run time -- bit-identical to the ahead-of-time generic form.
    Quiet form: the pretrained mask only; noisy form: any mask."""
    from bnn_chaos_model_amd import _native as N
    wa, w2, pd = (dev(swag_states[0][k][None]) for k in ("w_avg", "w2_avg", "pre_D"))
    for mask in (ops.V50_ZERO_MASK, 0):
        plan = N.Plan(mask)          # (a plan of its own: ops.get_plan's cached one may have had forms attached by the tests above)
        assert plan.v50net and not plan.spec_attached(False) and not plan.spec_attached(True)
        W = ops.swag_draw(wa, w2, pd, torch.zeros(3, dtype=torch.int32), philox_seed=5, plan=plan)
        for T in (99, 37, 6, 2, 100):
            x = dev(inputs["slow"][:, :T])
            for noisy in (False, True):
                kw = dict(philox_seed=4, draw_id0=9, system_id0=2, plan=plan, noisy=noisy, debug=True)
                a = ops.forward(x, W, engine="generic", **kw)
                if mask == 0 and not noisy:
                    with pytest.raises(N.NativeError):
                        ops.forward(x, W, engine="spec", **kw)      # no quiet form for another mask: an error, the default route stays generic
                    continue
                b = ops.forward(x, W, engine="spec", **kw)
                for u, v in zip(a, b):
                    assert torch.equal(u, v), (mask, T, noisy)
                if T != 100:
                    for u, v in zip(a, ops.forward(x, W, **kw)):
                        assert torch.equal(u, v)


def test_sharded_driver_specialize(ops):
    """distributed.MultiSwagSharded(specialize=True): the MC driver's moments through the network's specialised form == through the
    ahead-of-time form (same bits: float64 sums of identical samples)."""
    from bnn_chaos_model_amd.distributed import MultiSwagSharded
    arch = dict(hidden=24, latent=6, n_features=41, depth_in=1, depth_out=1)
    plan = ops.get_plan(**arch)
    assert not plan.spec_attached(False)
    g = torch.Generator(device="cuda").manual_seed(3)
    wa = torch.randn(2, plan.d, generator=g, device="cuda") * 0.25
    w2 = wa ** 2 + 1e-4
    pd = wa[:, :, None] + 0.01 * torch.randn(2, plan.d, 5, generator=g, device="cuda")
    x = torch.randn(200, 100, 41, generator=g, device="cuda")
    idx = torch.tensor([0, 1] * 8, dtype=torch.int32)
    a = MultiSwagSharded(wa, w2, pd, draws_per_launch=4, **arch).local_moments(x, idx, 7, 0)
    b = MultiSwagSharded(wa, w2, pd, draws_per_launch=4, specialize=True, **arch).local_moments(x, idx, 7, 0)
    assert plan.spec_attached(False) and not plan.spec_attached(True)
    assert a.dtype == torch.float64 and torch.equal(a, b)


def test_specialised_form_at_scale_invariances(ops):
    """200 000 systems x 30 draws (6e6 evaluations) of the (64, 16) network through its specialised form == the ahead-of-time form, bit
    for bit; and the size-independent properties of the path hold on it: no dependence on the block size, on system sharding (global
    Philox ids), on draw slabs; chunked draws cut torch.chunk pieces."""
    plan = _plan(ops, 64, 16, 1, 1)
    ops.specialize(plan, noisy=(False,))
    g = torch.Generator(device="cuda").manual_seed(21)
    B, J, seed = 200_000, 30, 5
    x = torch.randn(B, 100, 41, generator=g, device="cuda") * 0.1 + torch.randn(B, 1, 41, generator=g, device="cuda")
    W = torch.randn(J, plan.d, generator=g, device="cuda") * 0.2
    kw = dict(philox_seed=seed, draw_id0=60, system_id0=1000, plan=plan)
    full = ops.forward(x, W, engine="spec", **kw)
    assert torch.isfinite(full).all() and torch.equal(full, ops.forward(x, W, engine="generic", **kw))
    for spb in (64, 1024):
        assert torch.equal(full, ops.forward(x, W, engine="spec", systems_per_block=spb, **kw))
    cut = 70_001
    lo = ops.forward(x[:cut], W, engine="spec", **kw)
    hi = ops.forward(x[cut:], W, engine="spec", **dict(kw, system_id0=1000 + cut))
    assert torch.equal(full, torch.cat([lo, hi], 1))
    assert torch.equal(full[10:20], ops.forward(x, W[10:20], engine="spec", **dict(kw, draw_id0=70)))
    ch = ops.forward(x, W, engine="spec", nchunks=10, **kw)        # draw j covers chunk j % 10 of the systems (torch.chunk)
    assert ch.shape == (3, B, 2) and torch.equal(ch, ops.forward(x, W, engine="generic", nchunks=10, **kw))
    csz = -(-B // 10)
    for r, c in ((0, 0), (1, 3), (2, 9)):
        sl = slice(c * csz, min((c + 1) * csz, B))
        alone = ops.forward(x, W[r * 10 + c:r * 10 + c + 1], engine="spec", **dict(kw, draw_id0=6 + r))   # the same draw over every system, output-row id 60 / 10 + r
        assert torch.equal(ch[r, sl], alone[0, sl])


def random_archs():
    """[(hidden, latent, in, out, features, fix_megno, mask, T)] of the sweep below (seeded; scripts/prewarm_spec.py compiles them too)."""
    from bnn_chaos_model_amd.ops import V50_ZERO_MASK
    rng = np.random.default_rng(777)
    out = []
    for trial in range(7):
        F = 82 if trial == 4 else 41
        H = int(rng.choice([1, 8, 17, 24, 49, 64, 77, 100]))
        L = int(rng.choice([1, 2, 5, 12, 20, 31, 48]))
        din, dout = int(rng.integers(0, 4)), int(rng.integers(0, 3))
        megno = bool(rng.integers(0, 2))
        mask = (V50_ZERO_MASK if trial % 3 == 1 else int(rng.integers(0, 1 << 41))) | ((1 << 7) if megno else 0)
        out.append((H, L, din, dout, F, megno, mask, int(rng.choice([2, 3, 5, 8, 37, 100]))))
    return out


def test_random_architectures_every_candidate_form(ops):
    """Seeded random networks (widths 1..100, depths 0..3, 41 / 82 features, fix_megno, random column masks, ragged T): EVERY candidate
    form the specialiser would choose from (waves per workgroup x layer routine) equals the ahead-of-time generic engine bit for bit,
    quiet and noisy.  (scripts/dev/spec_random_sweep.py is the long version: 16 networks, 112 forms.)"""
    from bnn_chaos_model_amd import _native as N, specialize as S
    rng = np.random.default_rng(778)
    nforms = 0
    for (H, L, din, dout, F, megno, mask, T) in random_archs():
        try:
            plan = N.Plan(mask, 0.5, n_features=F, hidden=H, latent=L, fix_megno=megno, depth_in=din, depth_out=dout)
        except N.NativeError as e:      # outside the LDS budget: an error, never a wrong answer
            assert "LDS" in str(e), e
            continue
        x = dev((rng.standard_normal((37, 1, F)) + 0.2 * rng.standard_normal((37, T, F))).astype(np.float32))
        W = dev((rng.standard_normal((3, plan.d)) * (0.6 / np.sqrt(max(H, 8)))).astype(np.float32))
        for nz in (False, True):
            kw = dict(philox_seed=9, draw_id0=3, system_id0=11, plan=plan, noisy=nz, debug=True)
            ref = ops.forward(x, W, engine="generic", **kw)
            forms = S.candidates(plan.arch, nz)
            try:     # the resident-weights form (measured, not searched) where the builder offers it
                res = N.spec_source(plan.arch, nz, True, N.SPEC_POOL_REGS | N.SPEC_RESIDENT)
                forms.append((S.compile_source(res)[0], dict(w8=True, flags=N.SPEC_POOL_REGS | N.SPEC_RESIDENT)))
            except N.NativeError:
                pass
            for image, info in forms:
                plan.attach_spec(image, nz, info["w8"], info["flags"])
                for u, v in zip(ref, ops.forward(x, W, engine="spec", **kw)):
                    assert torch.equal(u, v), (H, L, din, dout, F, megno, hex(mask), T, nz, info)
                nforms += 1
    assert nforms >= 24
