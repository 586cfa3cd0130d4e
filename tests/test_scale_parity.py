"""Parity AT SCALE against the reference itself (tests/golden/case_scale.npz, made by make_golden_scale.py from the UNMODIFIED
/root/reference/spock_reg_model.py): all 30 pretrained members x 2 weight draws x 4 096 systems each through forward_swag_fast
(:878-908) and 30 x 512 systems through the noisy forward (:486-528) -- 261 120 evaluations, 522 240 outputs, each with the
float64 truth from the reference run in double.  Inputs and normals are NOT stored: tests/golden/scale_recipe.py regenerates
them bit for bit (integer hash + one IEEE rounding), checked against the fixture's CRC-32s.

BASELINE.json's bar is 1e-5 relative TO THE REFERENCE.  Measured (profiles/r06_scale_parity.json): 0 of the 522 240 outputs beyond it,
maximum 6.8e-6 -- the kernels keep the reference's per-output summation order (bias, inputs ascending) in every Linear layer, so the two
fp32 evaluations differ only in the pool's partitioning and in tanhf / expf.  Against the float64 TRUTH neither is within 1e-5 everywhere:
the reference misses it 28 times (max 1.84e-5), the HIP path 30 times (max 1.79e-5) -- the fp32 noise floor of this network, the same for
both.  The tests assert exactly that: no exceedance against the reference, and no farther from the truth than the reference is."""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden

sys.path.insert(0, GOLDEN)
import scale_recipe as R  # noqa: E402


def rel_stats(a, b):
    """a vs b, relative to |b|, mu and std separately: count beyond 1e-5, max, 99.9 %, median."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    out = {}
    for k, name in enumerate(("mu", "std")):
        r = np.abs(a[..., k] - b[..., k]) / np.abs(b[..., k])
        out[name] = {"n": int(r.size), "beyond_1e-5": int((r > 1e-5).sum()), "max": float(r.max()),
                     "p99.9": float(np.quantile(r, 0.999)), "median": float(np.median(r))}
    return out


@pytest.fixture(scope="module")
def case():
    z = load_golden("case_scale.npz")
    M, J, B, NB = (int(v) for v in z["shape"])
    assert (M, J, B, NB) == (R.MEMBERS, R.DRAWS, R.SYSTEMS, R.NOISY_SYSTEMS)
    truth = z["out32"].astype(np.float64) + z["truth_delta"].astype(np.float64)
    noisy_truth = z["noisy32"].astype(np.float64) + z["noisy_truth_delta"].astype(np.float64)
    return {"out32": z["out32"], "truth": truth, "noisy32": z["noisy32"], "noisy_truth": noisy_truth, "crc": z["crc"]}


def test_recipe_reproduces_the_generators_inputs(case):
    """The arrays the reference was fed, re-derived here: CRC-32 over x block + every noise array of three members."""
    for m in (0, 17, 29):
        assert R.checksums([m])[0] == case["crc"][m], m


def test_reference_against_its_own_float64_run(case):
    """What the REFERENCE does against the truth at this size -- the yardstick for everything else (numbers only; no GPU)."""
    s = rel_stats(case["out32"], case["truth"])
    assert s["mu"]["max"] < 5e-5 and s["std"]["max"] < 5e-5
    assert s["mu"]["beyond_1e-5"] + s["std"]["beyond_1e-5"] > 0, "the fp32 reference is NOT within 1e-5 of the truth everywhere: if this " \
        "ever fails the fixture changed"


def test_oracle_against_the_reference_at_scale(case):
    """The CPU restatement (natural summation order, fp32) on 30 members x 512 systems x 2 draws vs the reference's outputs, and its
    float64 build (first draw) vs the reference's float64 run: pins both the oracle's fp64 mode and the fixture's truth."""
    from oracle import oracle as orc
    z = load_golden("ensemble_v50.npz")
    ens = {k: z[k] for k in ("w_avg", "w2_avg", "pre_D")}    # (an NpzFile decompresses the member on EVERY access)
    nb = 512
    got32 = np.empty((R.MEMBERS, R.DRAWS, nb, 2), np.float32)
    got64 = np.empty((R.MEMBERS, nb, 2), np.float64)
    for m in range(R.MEMBERS):
        x = R.x_block(m, 0, nb)
        for j in range(R.DRAWS):
            z1, z2, e1, e2 = R.draw_noise(m, j, nb)
            w = orc.swag_draw(ens["w_avg"][m], ens["w2_avg"][m], ens["pre_D"][m], z1, z2, scale=0.5)
            got32[m, j] = orc.forward(x, w, e1, e2)
            if j == 0:
                w = orc.swag_draw(ens["w_avg"][m], ens["w2_avg"][m], ens["pre_D"][m], z1, z2, scale=0.5, dtype=np.float64)
                got64[m] = orc.forward(x, w, e1, e2, dtype=np.float64)
    s64 = rel_stats(got64, case["truth"][:, 0, :nb])
    assert max(s64["mu"]["max"], s64["std"]["max"]) < 1e-9, s64          # same math in double: agreement to rounding
    s32 = rel_stats(got32, case["out32"][:, :, :nb])
    assert max(s32["mu"]["max"], s32["std"]["max"]) < 5e-5, s32
    assert s32["mu"]["beyond_1e-5"] + s32["std"]["beyond_1e-5"] <= 6, s32  # of 61 440 outputs (measured: 0, max 6.0e-6 -- the oracle's natural order is close to MKL's)


# ------------------------------------------------------------------------------------------------------------------ GPU
def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_hip_against_the_reference_at_scale(case):
    """Every evaluation of the fixture through the HIP path's default route (ops.multiswag / ops.forward, explicit noise = the
    reference's normals): the exceedance count, maximum and 99.9 % of HIP-vs-reference, and of both against the float64 truth.
    Writes the table (gpurun_out/r06_scale_parity.json; the judged copy is profiles/r06_scale_parity.json) and holds the HIP path to
    the measured envelope."""
    import torch
    from bnn_chaos_model_amd import ops
    assert torch.cuda.is_available()
    ens = load_golden("ensemble_v50.npz")
    wa, w2, pd = _dev(ens["w_avg"]), _dev(ens["w2_avg"]), _dev(ens["pre_D"])
    M, J, B, NB = R.MEMBERS, R.DRAWS, R.SYSTEMS, R.NOISY_SYSTEMS
    hip = np.empty((M, J, B, 2), np.float32)
    hip_noisy = np.empty((M, NB, 2), np.float32)

    def inputs(m):
        assert R.checksums([m])[0] == case["crc"][m], f"recipe does not reproduce member {m}'s inputs"
        dr = [R.draw_noise(m, j) for j in range(J)]
        return R.x_block(m), dr, R.noisy_noise(m)

    with ThreadPoolExecutor(8) as ex:
        for m, (x, dr, nn) in enumerate(ex.map(inputs, range(M))):
            xg = _dev(x)
            z1 = _dev(np.concatenate([d[0] for d in dr], 0))                       # [J, d]
            z2 = _dev(np.stack([d[1][:, 0] for d in dr]))                           # [J, K]
            eps = _dev(np.stack([np.stack([d[2], d[3]], 1) for d in dr]))           # [J, B, 2, 20]
            idx = torch.full((J,), m, dtype=torch.int32)
            hip[m] = ops.multiswag(xg, wa, w2, pd, idx, z1, z2, eps, scale=0.5).cpu().numpy()
            e_in, e1, e2, e_sum = nn
            hip_noisy[m] = ops.forward(xg[:NB], wa[m][None], eps=_dev(np.stack([e1, e2], 1)[None]), eps_in=_dev(e_in[None]),
                                       eps_sum=_dev(e_sum[None])).cpu().numpy()[0]
    table = {
        "fixture": "tests/golden/case_scale.npz (the unmodified reference; make_golden_scale.py)",
        "evaluations": {"forward_swag_fast": M * J * B, "noisy_forward": M * NB},
        "forward_swag_fast": {"hip_vs_reference": rel_stats(hip, case["out32"]), "reference_vs_truth": rel_stats(case["out32"], case["truth"]),
                              "hip_vs_truth": rel_stats(hip, case["truth"])},
        "noisy_forward": {"hip_vs_reference": rel_stats(hip_noisy, case["noisy32"]),
                          "reference_vs_truth": rel_stats(case["noisy32"], case["noisy_truth"]),
                          "hip_vs_truth": rel_stats(hip_noisy, case["noisy_truth"])},
        "bit_identical_outputs": {"forward_swag_fast": int((hip == case["out32"]).sum()), "noisy_forward": int((hip_noisy == case["noisy32"]).sum())},
    }
    # which outputs miss the truth by more than 1e-5: the same ones for the reference and for the HIP path?
    beyond = lambda a: np.abs(a.astype(np.float64) - case["truth"]) > 1e-5 * np.abs(case["truth"])
    br, bh = beyond(case["out32"]), beyond(hip)
    table["forward_swag_fast"]["beyond_1e-5_of_truth"] = {"reference": int(br.sum()), "hip": int(bh.sum()), "both": int((br & bh).sum()),
                                                          "systems_involved": int((br | bh).any(-1).sum())}
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r06_scale_parity.json"), "w") as f:
            json.dump(table, f, indent=1)
    except OSError:
        pass
    print(json.dumps(table, indent=1))
    assert np.isfinite(hip).all() and np.isfinite(hip_noisy).all()
    for leg in ("forward_swag_fast", "noisy_forward"):
        t = table[leg]
        for k in ("mu", "std"):
            hr, rt, ht = t["hip_vs_reference"][k], t["reference_vs_truth"][k], t["hip_vs_truth"][k]
            # the envelope (ENVELOPE below): maxima of two fp32 evaluations in different summation orders, and of either against the truth
            assert hr["max"] <= ENVELOPE["max_between_fp32"], (leg, k, hr)
            assert ht["max"] <= ENVELOPE["max_vs_truth"], (leg, k, ht)
            assert hr["p99.9"] <= ENVELOPE["p999_between_fp32"], (leg, k, hr)
            # "no farther from the truth than the reference is", as distributions: the 99.9 % point within 25 %, the median within 25 %
            assert ht["p99.9"] <= 1.25 * rt["p99.9"] and ht["median"] <= 1.25 * rt["median"], (leg, k, ht, rt)
    tot = lambda leg, pair: sum(table[leg][pair][k]["beyond_1e-5"] for k in ("mu", "std"))
    assert tot("forward_swag_fast", "hip_vs_reference") <= ENVELOPE["beyond_between_fp32"], tot("forward_swag_fast", "hip_vs_reference")
    assert tot("forward_swag_fast", "hip_vs_truth") <= ENVELOPE["beyond_vs_truth"], tot("forward_swag_fast", "hip_vs_truth")


# Measured on the MI355X (profiles/r06_scale_parity.json): HIP vs the reference 0 of 522 240 outputs beyond 1e-5, max 6.8e-6, 99.9 % 2.1e-6
# (59 % of the outputs bit-identical); against the float64 truth the reference has 28 outputs beyond 1e-5 (max 1.84e-5) and the HIP path 30
# (max 1.79e-5).  So BASELINE's bar -- 1e-5 relative to the reference, NO exceedance -- holds at this size and is what is asserted; the
# truth-side numbers get the measured envelope with a little room for another compiler's tanhf / expf (the HIP path is bit-deterministic).
ENVELOPE = {"max_between_fp32": 1e-5, "beyond_between_fp32": 0, "p999_between_fp32": 3e-6, "max_vs_truth": 2.5e-5, "beyond_vs_truth": 40}


@pytest.mark.gpu
def test_module_surface_replays_the_reference_at_scale(case, tmp_path):
    """The same numbers through the DROP-IN SURFACE: SWAGModel.forward_swag_fast (model and x on the GPU, the reference's draws handed out
    by Player in the reference's order) for a negative-variance member, 4 096 systems: identical to the ops route bit for bit."""
    import torch
    from bnn_chaos_model_amd import checkpoint, ops
    from bnn_chaos_model_amd import spock_reg_model as srm
    ze = load_golden("ensemble_v50.npz")
    ens = {k: ze[k] for k in ("w_avg", "w2_avg", "pre_D")}
    z0 = load_golden("swag_v50_12.npz")
    m = 12
    path = str(tmp_path / "steps=300000_v50_12_output.pkl")
    checkpoint.write_swag_file(path, json.loads(str(z0["hparams_json"])), json.loads(str(z0["swa_params_json"])),
                               torch.tensor(ens["w_avg"][m]), torch.tensor(ens["w2_avg"][m]), torch.tensor(ens["pre_D"][m]))
    model = srm.load_swag(path).cuda().eval()
    x = _dev(R.x_block(m))
    z1, z2, e1, e2 = R.draw_noise(m, 1)
    with R.Player([z1, z2, e1, e2]):
        out = model.forward_swag_fast(x, scale=0.5)
    assert out.is_cuda and out.shape == (R.SYSTEMS, 2)
    direct = ops.multiswag(x, _dev(ens["w_avg"][m][None]), _dev(ens["w2_avg"][m][None]), _dev(ens["pre_D"][m][None]),
                           torch.zeros(1, dtype=torch.int32), _dev(z1), _dev(z2[:, 0][None]), _dev(np.stack([e1, e2], 1)[None]))[0]
    assert torch.equal(out, direct)
    s = rel_stats(out.cpu().numpy(), case["out32"][m, 1])
    assert max(s["mu"]["max"], s["std"]["max"]) <= ENVELOPE["max_between_fp32"], s


# ---------------------------------------------------------------------------------------- other hparams-built networks, at scale
ARCH_NAMES = ("h64l16", "h20l10", "h33l7", "deep22", "deep30", "lin00", "k40", "h48megno", "allcols", "lin0out8")


def _arch_hparams(z):
    hp = json.loads(str(z["hparams_json"]))
    for k, v in list(hp.items()):
        if isinstance(v, str) and v in ("True", "False"):
            hp[k] = v == "True"
    return hp


def _arch_kw(z):
    hp = _arch_hparams(z)
    flags = (hp.get("fix_megno", False), hp.get("fix_megno2", False), hp["include_mmr"], hp["include_nan"], hp.get("include_eplusminus", True))
    lowest = 0.1 if hp.get("lower_std", False) else 0.5
    net = dict(n_features=int(z["n_features"]), hidden=hp["hidden"], latent=hp["latent"], depth_in=hp["in"], depth_out=hp["out"])
    return flags, lowest, bool(hp.get("fix_megno", False)), net, int(json.loads(str(z["swa_params_json"]))["K"])


def test_oracle_against_the_reference_at_scale_other_networks():
    """case_scale_arch.npz (make_golden_scale_arch.py): the unmodified reference class built with OTHER hparams -- widths, depths, masks,
    K = 40, fix_megno: ten networks -- on 2 048 recipe systems each.  The oracle (natural order) on the first 256 of each vs the
    reference's outputs; its float64 build vs the reference's float64 run."""
    from oracle import oracle as orc
    zs = load_golden("case_scale_arch.npz")
    assert tuple(str(n) for n in zs["names"]) == ARCH_NAMES
    nb = 256
    worst32 = worst64 = 0.0
    nbad = 0
    for k, name in enumerate(ARCH_NAMES):
        z = load_golden(f"case_arch_{name}.npz")
        flags, lowest, megno, net, K = _arch_kw(z)
        mask = orc.zero_mask_from_flags(*flags)
        arch = orc.make_arch(T=100, zero_mask=mask, lowest=lowest, fix_megno=megno, **net)
        d, L = int(z["w_avg"].size), net["latent"]
        blk = int(zs["block0"]) + k
        x = R.x_block(blk, 0, nb)
        z1, z2, e1, e2 = R.draw_noise(blk, 0, nb, d=d, k=K, latent=L)
        ref = zs[f"{name}_fast32"][:nb]
        truth = ref.astype(np.float64) + zs[f"{name}_fast_truth_delta"][:nb]
        for dt in (np.float32, np.float64):
            w = orc.swag_draw(z["w_avg"], z["w2_avg"], z["pre_D"], z1, z2, scale=0.5, dtype=dt)
            got = orc.forward(x, w, e1, e2, arch=arch, dtype=dt)
            s = rel_stats(got, ref if dt == np.float32 else truth)
            if dt == np.float32:
                worst32 = max(worst32, s["mu"]["max"], s["std"]["max"])
                nbad += s["mu"]["beyond_1e-5"] + s["std"]["beyond_1e-5"]
            else:
                worst64 = max(worst64, s["mu"]["max"], s["std"]["max"])
    assert worst64 < 1e-9, worst64
    assert nbad == 0 and worst32 < 1e-5, (nbad, worst32)


@pytest.mark.gpu
def test_hip_against_the_reference_at_scale_other_networks():
    """The generic engine (ahead-of-time forms, and the pretrained shapes' own kernels for K = 40) on the ten other networks of
    case_scale_arch.npz: 2 048 systems through forward_swag_fast and 512 through the noisy forward each, against the reference's outputs
    -- no exceedance of 1e-5 -- and against its float64 run (no farther from the truth than the reference, as distributions).  Table:
    gpurun_out/r06_scale_parity_arch.json (judged copy: profiles/)."""
    import torch
    from bnn_chaos_model_amd import ops
    zs = load_golden("case_scale_arch.npz")
    S, NB = int(zs["systems"]), int(zs["noisy_systems"])
    table = {}
    tot_bad = 0
    import warnings
    for k, name in enumerate(ARCH_NAMES):
        z = load_golden(f"case_arch_{name}.npz")
        flags, lowest, megno, net, K = _arch_kw(z)
        plan = ops.get_plan(ops.zero_mask_from_flags(*flags), lowest, fix_megno=megno, **net)
        d, L = int(z["w_avg"].size), net["latent"]
        SM = 2 * L + (2 if megno else 0)
        blk = int(zs["block0"]) + k
        x = _dev(R.x_block(blk, 0, S))
        wa, w2, pd = _dev(z["w_avg"][None]), _dev(z["w2_avg"][None]), _dev(z["pre_D"][None])
        z1, z2, e1, e2 = R.draw_noise(blk, 0, S, d=d, k=K, latent=L)
        idx = torch.zeros(1, dtype=torch.int32)
        hip = ops.multiswag(x, wa, w2, pd, idx, _dev(z1), _dev(z2[:, 0][None]), _dev(np.stack([e1, e2], 1)[None]), scale=0.5, plan=plan)[0].cpu().numpy()
        e_in, n1, n2, e_sum = R.noisy_noise(blk, NB, latent=L, summary=SM)
        hipn = ops.forward(x[:NB], wa, eps=_dev(np.stack([n1, n2], 1)[None]), eps_in=_dev(e_in[None]), eps_sum=_dev(e_sum[None]), plan=plan)[0].cpu().numpy()
        row = {}
        for leg, got in (("fast", hip), ("noisy", hipn)):
            ref = zs[f"{name}_{leg}32"]
            truth = ref.astype(np.float64) + zs[f"{name}_{leg}_truth_delta"]
            row[leg] = {"hip_vs_reference": rel_stats(got, ref), "reference_vs_truth": rel_stats(ref, truth), "hip_vs_truth": rel_stats(got, truth),
                        "bit_identical": int((got == ref).sum())}
            for kk in ("mu", "std"):
                tot_bad += row[leg]["hip_vs_reference"][kk]["beyond_1e-5"]
                assert row[leg]["hip_vs_reference"][kk]["max"] <= 1e-5, (name, leg, kk, row[leg]["hip_vs_reference"][kk])
                assert row[leg]["hip_vs_truth"][kk]["max"] <= max(2.5e-5, 2.0 * row[leg]["reference_vs_truth"][kk]["max"]), (name, leg, kk)
        table[name] = row
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r06_scale_parity_arch.json"), "w") as f:
            json.dump(table, f, indent=1)
    except OSError:
        pass
    print(json.dumps({n: {leg: {"max_vs_ref": max(t[leg]["hip_vs_reference"][q]["max"] for q in ("mu", "std")), "bit_identical": t[leg]["bit_identical"]}
                          for leg in t} for n, t in table.items()}, indent=1))
    assert tot_bad == 0
