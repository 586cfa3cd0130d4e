"""Pins the CPU oracle's behaviour on NON-FINITE inputs to what the UNMODIFIED reference returns
(tests/golden/make_golden_nonfinite.py): `x = x - mask` turns NaN / +-inf in a MASKED column into NaN (spock_reg_model.py:452-478),
nn.ReLU propagates NaN and keeps +inf (:301-321), so such a system's (mu, std) is NaN -- except where an infinity meets weights of one
sign only and dies in the ReLU (the fixture's `dead` network), where the reference's outputs stay finite.  CPU only."""
import json

import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc


def tp(z, pfx):
    return [z[f"{pfx}_{i:03d}"] for i in range(int(z[pfx + "_n"]))]


def same_nan_close_elsewhere(got, want, rtol=1e-5):
    """NaN exactly where the reference has NaN (inf likewise, same sign); BASELINE.json's 1e-5 relative bar where it is finite."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want)), (np.argwhere(np.isnan(got) != np.isnan(want))[:8])
    inf = np.isinf(want)
    assert np.array_equal(got[inf], want[inf])
    fin = np.isfinite(want)
    err = np.abs(got[fin] - want[fin])
    assert (err <= rtol * np.abs(want[fin])).all(), err.max()


def dead_arch(z):
    hp = json.loads(str(z["dead_hparams_json"]))
    return orc.make_arch(T=100, hidden=int(hp["hidden"]), latent=int(hp["latent"]), depth_in=int(hp["in"]), depth_out=int(hp["out"]))


@pytest.fixture(scope="module")
def z():
    return load_golden("case_nonfinite.npz")


def test_the_fixture_covers_what_it_says(z):
    what = [str(w) for w in z["what"]]
    assert len(what) == z["x"].shape[0] == 20 and what[0] == "clean"
    x = z["x"]
    assert np.isfinite(x[0]).all() and np.isfinite(x[19]).all() and all(not np.isfinite(x[b]).all() for b in range(1, 19))
    for pfx in ("v50_0", "v50_12"):
        for call in ("swagfast", "forward_noisy0", "forward_noisy1"):
            out = z[f"{pfx}_{call}_out"]
            assert np.isfinite(out[0]).all() and np.isnan(out[1:]).all()     # every damaged system -- and the 1e30 one: its pool overflows
    d = z["dead_swagfast_out"]
    assert np.isfinite(d[[0, 1, 3, 5, 7]]).all() and np.isnan(d[[2, 4, 6]]).all()


@pytest.mark.parametrize("pfx", ("v50_0", "v50_12"))
@pytest.mark.parametrize("pool_parts", (1, 4))
def test_pretrained_members_on_damaged_systems(z, pfx, pool_parts):
    x = z["x"]
    sched = orc.make_schedule(None, pool_parts=pool_parts)
    t = tp(z, f"{pfx}_swagfast_tape")
    out, ex = orc.forward(x, z[f"{pfx}_swagfast_w"], t[2], t[3], sched=sched, extras=True)
    same_nan_close_elsewhere(out, z[f"{pfx}_swagfast_out"])
    w = z[f"{pfx}_swagfast_w"]
    t = tp(z, f"{pfx}_forward_noisy0_tape")
    out, ex = orc.forward(x, w, t[0], t[1], sched=sched, extras=True)
    same_nan_close_elsewhere(out, z[f"{pfx}_forward_noisy0_out"])
    # the side effects the reference leaves behind: _cur_summary (:512) and latents (:433) carry the same NaN / inf pattern
    want = z[f"{pfx}_forward_noisy0_summary"]
    assert np.array_equal(np.isnan(ex["summary"]), np.isnan(want))
    wl = z[f"{pfx}_forward_noisy0_latents"]
    assert np.array_equal(np.isnan(ex["latents"][:4]), np.isnan(wl)) and np.array_equal(np.isinf(ex["latents"][:4]), np.isinf(wl))
    t = tp(z, f"{pfx}_forward_noisy1_tape")
    out = orc.forward(x, w, t[1], t[2], eps_in=t[0], eps_sum=t[3], sched=sched)
    same_nan_close_elsewhere(out, z[f"{pfx}_forward_noisy1_out"])


def test_sample_on_damaged_systems(z):
    """VarModel.sample (:530-545): np.average over mu + randn * std -- NaN for every damaged system."""
    t = tp(z, "sample_tape")
    assert len(t) == 10
    acc = []
    for s in range(2):
        e_in, e1, e2, e_sum, nz = t[5 * s: 5 * s + 5]
        out = orc.forward(z["x"], z["sample_w"], e1, e2, eps_in=e_in, eps_sum=e_sum)
        acc.append(out[:, 0].astype(np.float64) + nz * out[:, 1].astype(np.float64))
    same_nan_close_elsewhere(np.average(acc, axis=0), z["sample_out"], rtol=2e-5)


@pytest.mark.parametrize("pool_parts", (1, 4))
def test_an_infinity_that_dies_in_the_relu_leaves_finite_outputs(z, pool_parts):
    """The `dead` network: +inf on a live column whose feature_nn.0 weights are all negative -> -inf -> ReLU -> 0: finite (mu, std),
    within 1e-5 of the reference; -inf on it, NaN, or a non-finite value in a masked column next to it: NaN."""
    arch = dead_arch(z)
    sched = orc.make_schedule(None, pool_parts=pool_parts)
    x = z["dead_x"]
    t = tp(z, "dead_swagfast_tape")
    w = orc.swag_draw(z["dead_w_avg"], z["dead_w2_avg"], z["dead_pre_D"], t[0], t[1], scale=0.5)
    assert np.abs(w.astype(np.float64) - z["dead_swagfast_w"]).max() <= 2e-6
    col = int(z["dead_col"])
    W1 = w[41 + 20: 41 + 20 + 20 * 41].reshape(20, 41)
    assert (W1[:, col] < 0).all()
    out = orc.forward(x, z["dead_swagfast_w"], t[2], t[3], arch=arch, sched=sched)
    same_nan_close_elsewhere(out, z["dead_swagfast_out"])
    assert np.isfinite(out[[1, 3, 5]]).all()
    t = tp(z, "dead_forward_noisy0_tape")
    out, ex = orc.forward(x, z["dead_swagfast_w"], t[0], t[1], arch=arch, sched=sched, extras=True)
    same_nan_close_elsewhere(out, z["dead_forward_noisy0_out"])
    want = z["dead_forward_noisy0_summary"]
    assert np.array_equal(np.isnan(ex["summary"]), np.isnan(want))
    t = tp(z, "dead_forward_noisy1_tape")
    out = orc.forward(x, z["dead_swagfast_w"], t[1], t[2], eps_in=t[0], eps_sum=t[3], arch=arch, sched=sched)
    same_nan_close_elsewhere(out, z["dead_forward_noisy1_out"])


def test_float64_restatement_agrees_on_the_pattern(z):
    """orc64_ on the same inputs: the same NaN pattern for genuinely non-finite inputs (the 1e30 system is finite in float64: its pool
    does not overflow there -- an fp32 effect the fp32 restatement shares with the reference)."""
    t = tp(z, "v50_0_swagfast_tape")
    out = orc.forward(z["x"], z["v50_0_swagfast_w"], t[2], t[3], dtype=np.float64)
    assert np.isfinite(out[[0, 19]]).all() and np.isnan(out[1:19]).all()
