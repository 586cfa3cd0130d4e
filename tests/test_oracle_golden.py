"""Pins the CPU oracle (oracle/bnn_oracle.c) to vectors produced by the UNMODIFIED reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import close_report, load_golden, tape
from oracle import oracle as orc

SEEDS = (0, 12)
INPUTS = ("slow", "iid", "const4")


@pytest.mark.parametrize("si", SEEDS)
def test_swag_draw_matches_reference(si, swag_states):
    st = swag_states[si]
    z = load_golden(f"case_swagfast_v50_{si}_slow.npz")
    tp = tape(z)
    w = orc.swag_draw(st["w_avg"], st["w2_avg"], st["pre_D"], tp[0][1], tp[1][1], scale=0.5)
    # diag term is bit-identical; the [d,K] GEMV differs from MKL only in summation order
    err = np.abs(w.astype(np.float64) - z["w"])
    assert err.max() <= 2e-6, err.max()
    assert np.mean(w == z["w"]) > 0.5
    # fp64 restatement against the fp64 truth of the fixture
    w64 = orc.swag_draw(st["w_avg"], st["w2_avg"], st["pre_D"], tp[0][1], tp[1][1], scale=0.5, dtype=np.float64)
    assert np.abs(w64 - z["w_f64"]).max() < 1e-12


def test_negative_variance_element_needs_abs(swag_states):
    st = swag_states[12]  # v50_12 has one negative w2_avg - w_avg^2
    var = st["w2_avg"] - st["w_avg"] ** 2
    assert (var < 0).sum() >= 1
    z = load_golden("case_swagfast_v50_12_slow.npz")
    assert np.isfinite(z["w"]).all()


@pytest.mark.parametrize("si", SEEDS)
@pytest.mark.parametrize("xname", INPUTS)
def test_forward_swag_fast_matches_reference(si, xname, swag_states, inputs):
    z = load_golden(f"case_swagfast_v50_{si}_{xname}.npz")
    tp = tape(z)
    assert [k for k, _ in tp] == ["torch.randn", "torch.randn", "torch.randn_like", "torch.randn_like"]
    x = inputs[xname]
    # forward with the REFERENCE's sampled weights isolates the forward restatement
    out, ex = orc.forward(x, z["w"], tp[2][1], tp[3][1], extras=True)
    nbad, mx = close_report(out, z["out"])
    assert nbad == 0, (nbad, mx)
    nbad, mx = close_report(ex["summary"], z["summary"], rtol=2e-5, atol=2e-5)
    assert nbad == 0, (nbad, mx)
    nbad, mx = close_report(ex["pre_clamp"], z["pre_clamp"], rtol=1e-4, atol=1e-4)
    assert nbad == 0, (nbad, mx)
    nbad, mx = close_report(ex["latents"][:2], z["latents"], rtol=1e-5, atol=2e-5)
    assert nbad == 0, (nbad, mx)
    # end to end: oracle draw + oracle forward
    st = swag_states[si]
    w = orc.swag_draw(st["w_avg"], st["w2_avg"], st["pre_D"], tp[0][1], tp[1][1], scale=0.5)
    out2 = orc.forward(x, w, tp[2][1], tp[3][1])
    nbad, mx = close_report(out2, z["out"])
    assert nbad == 0, (nbad, mx)
    # fp64 restatement vs the fixture's fp64 truth (same noise)
    out64 = orc.forward(x, z["w_f64"], tp[2][1], tp[3][1], dtype=np.float64)
    assert np.abs(out64 - z["out_f64"]).max() < 1e-9


@pytest.mark.parametrize("si", SEEDS)
@pytest.mark.parametrize("noisy", (0, 1))
def test_varmodel_forward_matches_reference(si, noisy, inputs):
    z = load_golden(f"case_forward_v50_{si}_noisy{noisy}.npz")
    tp = tape(z)
    x = inputs["slow"]
    if noisy:
        assert [a.shape for _, a in tp] == [(32, 100, 41), (32, 20), (32, 20), (32, 40)]
        out = orc.forward(x, z["w"], tp[1][1], tp[2][1], eps_in=tp[0][1], eps_sum=tp[3][1])
    else:
        assert [a.shape for _, a in tp] == [(32, 20), (32, 20)]
        out = orc.forward(x, z["w"], tp[0][1], tp[1][1])
    nbad, mx = close_report(out, z["out"])
    assert nbad == 0, (nbad, mx)


def test_varmodel_sample_matches_reference(inputs):
    """VarModel.sample (spock_reg_model.py:530-545): mean over samples of mu + randn*std."""
    z = load_golden("case_sample_v50_0.npz")
    tp = tape(z)
    x = inputs["slow"]
    n = int(z["samples"])
    assert len(tp) == 5 * n
    acc = []
    for s in range(n):
        e_in, e1, e2, e_sum, nz = (tp[5 * s + i][1] for i in range(5))
        out = orc.forward(x, z["w"], e1, e2, eps_in=e_in, eps_sum=e_sum)
        acc.append(out[:, 0].astype(np.float64) + nz * out[:, 1].astype(np.float64))
    got = np.average(acc, axis=0)
    nbad, mx = close_report(got, z["out"], rtol=2e-5, atol=2e-5)
    assert nbad == 0, (nbad, mx)


def _grid_from_tape(tp, B, L=20):
    """Unpack [randint, randn(1,d), randn(K,1), randn_like(Bc,L) x2] * J into arrays."""
    seed_idx, z1, z2, e1, e2 = [], [], [], [], []
    for i in range(0, len(tp), 5):
        assert tp[i][0] == "np.randint"
        seed_idx.append(int(tp[i][1]))
        z1.append(tp[i + 1][1].reshape(-1))
        z2.append(tp[i + 2][1].reshape(-1))
        e1.append(tp[i + 3][1])
        e2.append(tp[i + 4][1])
    return np.array(seed_idx, np.int32), np.stack(z1), np.stack(z2), e1, e2


def _stack_states(swag_states, members):
    return (np.stack([swag_states[m]["w_avg"] for m in members]), np.stack([swag_states[m]["w2_avg"] for m in members]),
            np.stack([swag_states[m]["pre_D"] for m in members]))


def test_multiswag_grid_matches_reference(swag_states, inputs):
    """sample_full_swag x3 (figures/spock/regression.py:74-92 semantics): seed pick + forward_swag_fast."""
    z = load_golden("case_multiswag_grid.npz")
    tp = tape(z)
    x = inputs["slow"]
    seed_idx, z1, z2, e1, e2 = _grid_from_tape(tp, x.shape[0])
    assert seed_idx.tolist() == [int(t[1]) for t in tp[::5]]
    eps = np.stack([np.stack([a, b], axis=1) for a, b in zip(e1, e2)])  # [J,B,2,L]
    wa, w2, pd = _stack_states(swag_states, z["ensemble"])
    out = orc.multiswag(x, wa, w2, pd, seed_idx, z1, z2, eps, nchunks=1)
    nbad, mx = close_report(out, z["out"])
    assert nbad == 0, (nbad, mx)


def test_chunk_loop_matches_reference(swag_states, inputs):
    """The 5-planet MC loop (figures/multiswag_5_planet.py:295-298): samples x torch.chunk(X, 10)."""
    z = load_golden("case_chunk_loop.npz")
    tp = tape(z)
    B, nch, S = int(z["nrows"]), int(z["chunks"]), int(z["samples"])
    x = inputs["slow"][:B]
    seed_idx, z1, z2, e1, e2 = _grid_from_tape(tp, B)
    assert len(seed_idx) == S * nch
    csz = -(-B // nch)
    eps = np.zeros((S, B, 2, 20), np.float32)
    for e in range(S * nch):
        s, c = divmod(e, nch)
        eps[s, c * csz:(c + 1) * csz, 0] = e1[e]
        eps[s, c * csz:(c + 1) * csz, 1] = e2[e]
    wa, w2, pd = _stack_states(swag_states, z["ensemble"])
    out = orc.multiswag(x, wa, w2, pd, seed_idx, z1, z2, eps, nchunks=nch)
    nbad, mx = close_report(out, z["out"])
    assert nbad == 0, (nbad, mx)


def test_schedule_changes_only_rounding(swag_states, inputs):
    """A pinned accumulation order (orc_schedule) must stay within rounding of the natural order."""
    z = load_golden("case_swagfast_v50_0_slow.npz")
    tp = tape(z)
    x = inputs["slow"]
    rng = np.random.default_rng(0)
    orders = [np.concatenate([rng.permutation(41), [-1]]), rng.permutation(40), rng.permutation(40),
              rng.permutation(40), rng.permutation(40), rng.permutation(40)]
    sched = orc.make_schedule(orders, pool_parts=4)
    a = orc.forward(x, z["w"], tp[2][1], tp[3][1])
    b = orc.forward(x, z["w"], tp[2][1], tp[3][1], sched=sched)
    nbad, mx = close_report(b, a)
    assert nbad == 0 and mx > 0, (nbad, mx)
    nbad, mx = close_report(b, z["out"])
    assert nbad == 0, (nbad, mx)


def test_known_answer_properties(swag_states, inputs):
    st = swag_states[0]
    z = load_golden("case_swagfast_v50_0_slow.npz")
    tp = tape(z)
    x = inputs["slow"].copy()
    # scale = 0 => w == w_avg exactly
    w0 = orc.swag_draw(st["w_avg"], st["w2_avg"], st["pre_D"], tp[0][1], tp[1][1], scale=0.0)
    assert np.array_equal(w0, st["w_avg"])
    base = orc.forward(x, z["w"], tp[2][1], tp[3][1])
    # masked columns do not affect the output
    x2 = x.copy()
    x2[:, :, [1, 2, 3, 4, 5, 6, 7, 38, 39, 40]] = 123.0
    assert np.array_equal(orc.forward(x2, z["w"], tp[2][1], tp[3][1]), base)
    # ranges of soft_clamp
    assert (base[:, 0] >= 4).all() and (base[:, 0] <= 12).all() and (base[:, 1] >= 0.5).all() and (base[:, 1] <= 6).all()
    # time-pool is order invariant up to rounding
    perm = np.random.default_rng(1).permutation(100)
    nbad, mx = close_report(orc.forward(x[:, perm], z["w"], tp[2][1], tp[3][1]), base)
    assert nbad == 0, (nbad, mx)


def test_feature_packing_matches_reference_function():
    """oracle/features.py vs data_setup_kernel executed from the reference's own source (+ ssX.transform + .float())."""
    from oracle import features
    z = load_golden("case_features.npz")
    X = np.stack([features.data_setup(z["mass"][i], z["tseries"][i][None])[0] for i in range(z["mass"].shape[0])])
    assert X.shape == z["X64"].shape and np.array_equal(X, z["X64"])
    assert np.array_equal(features.standardize(X, z["mean"], z["scale"]), z["x32"])
    assert np.isfinite(z["X64"]).all() and (z["X64"][0, 5, 38] == 1.0) and (z["X64"][1, 7, 39] == 1.0) and (z["X64"][2, 9, 40] == 1.0)


def test_statistics_epilogue_matches_reference_fragments():
    """oracle/stats.py vs fast_truncnorm, the prior-resampling block and the min over trios executed from the reference's source."""
    from oracle import stats
    z = load_golden("case_stats.npz")
    n = z["loc"].size
    d, nsamp = int(z["d"]), int(z["nsamp"])
    normals = np.concatenate([z[f"normals_{i:03d}"] for i in range(int(z["normals_n"]))], axis=1)  # chunks of d elements
    assert normals.shape == (nsamp, n)
    tn = stats.truncnorm_first_good(z["loc"], z["scale"], normals, float(z["left"]))
    assert tn.dtype == np.float32 and np.array_equal(tn, z["truncnorm"])
    assert (tn[0, :5] < 4).all()  # nothing passed: the first candidate is kept (argmax of an all-False mask)
    cum, edges = stats.prior_table(int(z["n_samples"]), float(z["normalization"]))
    assert np.array_equal(cum, z["cum_values"]) and np.array_equal(edges, z["bin_edges"])
    rs = stats.resample_prior(z["truncnorm"], z["u"], float(z["normalization"]))
    assert np.array_equal(rs, z["resampled"])
    assert np.array_equal(np.min(rs, 2).T, z["outs"])
    from scipy.integrate import quad
    assert quad(stats.prior_pdf, a=9, b=np.inf)[0] == float(z["normalization"])


def test_all_thirty_pretrained_seeds_match_reference():
    """Every member of the pretrained ensemble (tests/golden/ensemble_v50.npz, converted states) under the reference's own
    forward_swag_fast run (case_all_seeds.npz): oracle draw <= 2e-6 from the reference's sampled weights, outputs within the
    1e-5 relative bar.  The five members with a negative w2_avg - w_avg^2 element are among them."""
    ens = load_golden("ensemble_v50.npz")
    z = load_golden("case_all_seeds.npz")
    assert ens["w_avg"].shape == (30, 7583) and ens["pre_D"].shape == (30, 7583, 30)
    neg = np.flatnonzero(ens["negative_variance_elements"])
    assert neg.tolist() == [3, 12, 22, 25, 26]
    for i in range(30):
        var = ens["w2_avg"][i] - ens["w_avg"][i] ** 2
        assert int((var < 0).sum()) == int(ens["negative_variance_elements"][i])
        w = orc.swag_draw(ens["w_avg"][i], ens["w2_avg"][i], ens["pre_D"][i], z["z1"][i], z["z2"][i], scale=0.5)
        assert np.isfinite(w).all() and np.abs(w.astype(np.float64) - z["w"][i]).max() <= 2e-6, i
        out = orc.forward(z["x"], w, z["eps"][i, :, 0], z["eps"][i, :, 1])
        nbad, mx = close_report(out, z["out"][i])
        assert nbad == 0, (i, nbad, mx)
    # the two single-seed fixtures are rows of the ensemble
    for si in (0, 12):
        st = load_golden(f"swag_v50_{si}.npz")
        assert np.array_equal(st["w_avg"], ens["w_avg"][si]) and np.array_equal(st["pre_D"], ens["pre_D"][si])


def test_truncnorm_two_sided_and_right_sided_forms():
    """oracle/stats.truncnorm_first_good vs the reference's fast_truncnorm source for its other two acceptance tests."""
    from oracle import stats
    z = load_golden("case_truncnorm2.npz")
    for name in ("two_sided", "right_only"):
        got = stats.truncnorm_first_good(z["loc"], z["scale"], z[f"{name}_normals"], float(z[f"{name}_left"]), float(z[f"{name}_right"]))
        assert got.dtype == np.float32 and np.array_equal(got, z[f"{name}_out"]), name
    assert (z["two_sided_out"][0, :4] > 9).all()      # nothing inside (4, 9): the first candidate came back


def _tp(z, pfx):
    return [z[f"{pfx}_{i:03d}"] for i in range(int(z[pfx + "_n"]))]


def test_fix_megno_branch_matches_reference():
    """hparams['fix_megno'] = True (spock_reg_model.py:360-362, 480-491, 509-510): no pretrained checkpoint uses it, so the fixture
    (tests/golden/make_golden_megno.py) runs the unmodified reference class with that flag on a synthetic SWAG state.  d = 7665, the
    summary is 42 wide, its last two entries are the time mean / unbiased std of the RAW MEGNO column."""
    z = load_golden("case_megno.npz")
    arch = orc.make_arch(T=100, fix_megno=True)
    assert z["w_avg"].shape == (7665,) and (arch.zero_mask >> 7) & 1
    assert dict(zip(z["state_keys"].tolist(), z["state_sizes"].tolist()))["regress_nn.0.weight"] == 40 * 42
    t = _tp(z, "swagfast_tape")
    w = orc.swag_draw(z["w_avg"], z["w2_avg"], z["pre_D"], t[0], t[1])
    assert np.abs(w - z["swagfast_w"]).max() <= 2e-6
    nbad, mx = close_report(orc.forward(z["x"], w, t[2], t[3], arch=arch), z["swagfast_out"])
    assert nbad == 0, (nbad, mx)
    for noisy in (0, 1):
        t = _tp(z, f"forward_noisy{noisy}_tape")
        assert [a.shape for a in t] == ([(16, 100, 41)] if noisy else []) + [(16, 20), (16, 20)] + ([(16, 42)] if noisy else [])
        kw = dict(eps_in=t[0], eps_sum=t[3]) if noisy else {}
        e1, e2 = (t[1], t[2]) if noisy else (t[0], t[1])
        out, ex = orc.forward(z["x"], z["swagfast_w"], e1, e2, arch=arch, extras=True, **kw)
        nbad, mx = close_report(out, z[f"forward_noisy{noisy}_out"])
        assert nbad == 0, (noisy, nbad, mx)
        ref = z[f"forward_noisy{noisy}_summary"]
        assert ex["summary"].shape == ref.shape == (16, 42)
        assert np.abs(ex["summary"][:, 40:] - ref[:, 40:]).max() <= 1e-5 * np.abs(ref[:, 40:]).max()
        if not noisy:   # the MEGNO statistics are those of the raw column, whatever the mask does to it afterwards
            assert np.allclose(ref[:, 40], z["x"][:, :, 7].mean(1), rtol=1e-6) and np.allclose(ref[:, 41], z["x"][:, :, 7].std(1, ddof=1), rtol=1e-5)
            nbad, mx = close_report(ex["summary"][:, :40], ref[:, :40], rtol=2e-5, atol=2e-5)
            assert nbad == 0, (nbad, mx)
    # the pool schedule of the GPU kernels (4 strided partitions, pairwise merge) changes rounding only
    sch = orc.make_schedule(None, pool_parts=4)
    t = _tp(z, "forward_noisy0_tape")
    o4 = orc.forward(z["x"], z["swagfast_w"], t[0], t[1], arch=arch, sched=sch)
    nbad, mx = close_report(o4, z["forward_noisy0_out"])
    assert nbad == 0, (nbad, mx)


def test_forward_with_random_sample_matches_reference():
    """VarModel.forward with random_sample = True (`augment`, spock_reg_model.py:404-408, :502-503): the timesteps the reference picked
    (taped), behind the masks and in front of the noise -- the oracle on x[:, picked] at that series length."""
    z = load_golden("case_augment.npz")
    x = z["x"]
    for i in range(int(z["runs"])):
        t = [z[f"run{i}_tape_{j:03d}"] for j in range(int(z[f"run{i}_tape_n"]))]
        n_t, idx = int(t[0]), np.asarray(t[1])
        assert idx.shape == (n_t,) and 5 <= n_t <= 100 and idx.min() >= 0 and idx.max() < 100
        xa = np.ascontiguousarray(x[:, idx])
        arch = orc.make_arch(T=n_t)
        if int(z[f"run{i}_noisy"]):
            assert t[2].shape == (16, n_t, 41)     # the input noise has the augmented shape
            out = orc.forward(xa, z["w"], t[3], t[4], eps_in=t[2], eps_sum=t[5], arch=arch)
        else:
            out = orc.forward(xa, z["w"], t[2], t[3], arch=arch)
        nbad, mx = close_report(out, z[f"run{i}_out"])
        assert nbad == 0, (i, nbad, mx)
