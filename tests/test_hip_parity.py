"""HIP kernels (through the C ABI) vs the CPU oracle and the reference-generated golden vectors.
Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

from conftest import close_report, load_golden, tape

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU box"
    from bnn_chaos_model_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as _orc
    return _orc


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def kernel_schedule(ops, orc, noisy=False):
    """The accumulation order the kernels use, handed to the oracle so results can be compared bit for bit."""
    plan = ops.get_plan()
    return orc.make_schedule([plan.layer_order(l, noisy) for l in range(6)], pool_parts=4)


def stack_states(swag_states, members):
    return (np.stack([swag_states[m]["w_avg"] for m in members]), np.stack([swag_states[m]["w2_avg"] for m in members]),
            np.stack([swag_states[m]["pre_D"] for m in members]))


def test_library_loads_and_sees_gpu(ops):
    from bnn_chaos_model_amd import _native as N
    assert N.lib().bnn_device_count() >= 1
    assert ops.get_plan().d == 7583


def test_philox_known_answer(ops):
    # Random123 kat_vectors: philox4x32-10, counter 0, key 0 / all ones / pi digits
    r = ops.philox_raw((0, 0, 0, 0), (0, 0), 2).cpu().numpy().view(np.uint32)
    assert [hex(v) for v in r[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    r = ops.philox_raw((0xffffffff,) * 4, (0xffffffff, 0xffffffff), 1).cpu().numpy().view(np.uint32)
    assert [hex(v) for v in r[0]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    r = ops.philox_raw((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), 1).cpu().numpy().view(np.uint32)
    assert [hex(v) for v in r[0]] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


@pytest.mark.parametrize("si", (0, 12))
def test_swag_draw_bit_exact_vs_oracle(si, ops, orc, swag_states):
    st = swag_states[si]
    z = load_golden(f"case_swagfast_v50_{si}_slow.npz")
    tp = tape(z)
    W = ops.swag_draw(dev(st["w_avg"][None]), dev(st["w2_avg"][None]), dev(st["pre_D"][None]),
                      torch.zeros(1, dtype=torch.int32), dev(tp[0][1].reshape(1, -1)), dev(tp[1][1].reshape(1, -1)), scale=0.5)
    w = W.cpu().numpy()[0]
    w_orc = orc.swag_draw(st["w_avg"], st["w2_avg"], st["pre_D"], tp[0][1], tp[1][1], scale=0.5)
    assert np.array_equal(w, w_orc), np.abs(w - w_orc).max()
    assert np.abs(w.astype(np.float64) - z["w"]).max() <= 2e-6  # vs the reference's own draw


@pytest.mark.parametrize("si", (0, 12))
@pytest.mark.parametrize("xname", ("slow", "iid", "const4"))
def test_forward_vs_reference_and_oracle(si, xname, ops, orc, inputs):
    z = load_golden(f"case_swagfast_v50_{si}_{xname}.npz")
    tp = tape(z)
    x = inputs[xname]
    B = x.shape[0]
    eps = np.stack([tp[2][1], tp[3][1]], axis=1)[None]  # [1,B,2,20]
    out, pre, summ = ops.forward(dev(x), dev(z["w"][None]), eps=dev(eps), debug=True)
    out, pre, summ = out.cpu().numpy()[0], pre.cpu().numpy()[0], summ.cpu().numpy()[0]
    # 1) the reference's own outputs (tolerance of BASELINE.json: 1e-5 relative fp32)
    nbad, mx = close_report(out, z["out"])
    assert nbad == 0, (nbad, mx)
    # 2) the oracle in its natural order
    o_nat = orc.forward(x, z["w"], tp[2][1], tp[3][1])
    nbad, mx = close_report(out, o_nat)
    assert nbad == 0, (nbad, mx)
    # 3) the oracle pinned to the kernels' accumulation order: bit for bit up to tanh
    o_k, ex = orc.forward(x, z["w"], tp[2][1], tp[3][1], sched=kernel_schedule(ops, orc), extras=True)
    assert np.array_equal(summ, ex["summary"]), np.abs(summ - ex["summary"]).max()
    assert np.array_equal(pre, ex["pre_clamp"]), np.abs(pre - ex["pre_clamp"]).max()
    assert np.abs(out - o_k).max() <= 2e-6


@pytest.mark.parametrize("si", (0, 12))
def test_regress_is_the_tail_of_forward(si, ops, inputs):
    """bnn_regress_f32 (predict_instability, :437-442) on the summary bnn_forward_f32 reports == that forward's outputs, bit for bit."""
    z = load_golden(f"case_swagfast_v50_{si}_slow.npz")
    tp = tape(z)
    eps = np.stack([tp[2][1], tp[3][1]], axis=1)[None]
    W = dev(np.stack([z["w"], z["w"] * 0.5]))
    out, pre, summ = ops.forward(dev(inputs["slow"]), W, eps=dev(np.concatenate([eps, eps])), debug=True)
    out2, pre2 = ops.regress(summ, W, debug=True)
    assert torch.equal(pre2, pre) and torch.equal(out2, out)
    nbad, mx = close_report(out2[0].cpu().numpy(), z["out"])
    assert nbad == 0, (nbad, mx)
    # ragged system counts, empty batch
    for B in (1, 127, 129):
        o = ops.regress(summ[:, :B].contiguous(), W)
        assert torch.equal(o, out[:, :B])
    assert ops.regress(summ[:, :0].contiguous(), W).shape == (2, 0, 2)
    # more draws than one grid dimension holds (launches are split at 65535 draws)
    Jb = 65535 + 3
    big = ops.regress(summ[0, :1].expand(Jb, 1, 40).contiguous(), W[:1].expand(Jb, -1).contiguous())
    assert torch.equal(big, out[0, :1].expand(Jb, 1, 2))
    with pytest.raises(ValueError):
        ops.regress(summ, W[:1])


@pytest.mark.parametrize("si", (0, 12))
@pytest.mark.parametrize("noisy", (0, 1))
def test_varmodel_forward_vs_reference(si, noisy, ops, orc, inputs):
    z = load_golden(f"case_forward_v50_{si}_noisy{noisy}.npz")
    tp = tape(z)
    x = inputs["slow"]
    if noisy:
        e_in, e1, e2, e_sum = (t[1] for t in tp)
        eps = np.stack([e1, e2], axis=1)[None]
        out = ops.forward(dev(x), dev(z["w"][None]), eps=dev(eps), eps_in=dev(e_in[None]), eps_sum=dev(e_sum[None]))
        ref_o = orc.forward(x, z["w"], e1, e2, eps_in=e_in, eps_sum=e_sum)
    else:
        e1, e2 = tp[0][1], tp[1][1]
        eps = np.stack([e1, e2], axis=1)[None]
        out = ops.forward(dev(x), dev(z["w"][None]), eps=dev(eps))
        ref_o = orc.forward(x, z["w"], e1, e2)
    out = out.cpu().numpy()[0]
    nbad, mx = close_report(out, z["out"])
    assert nbad == 0, (nbad, mx)
    nbad, mx = close_report(out, ref_o)
    assert nbad == 0, (nbad, mx)


def _grid_from_tape(tp):
    seed_idx, z1, z2, e1, e2 = [], [], [], [], []
    for i in range(0, len(tp), 5):
        seed_idx.append(int(tp[i][1]))
        z1.append(tp[i + 1][1].reshape(-1))
        z2.append(tp[i + 2][1].reshape(-1))
        e1.append(tp[i + 3][1])
        e2.append(tp[i + 4][1])
    return np.array(seed_idx, np.int32), np.stack(z1), np.stack(z2), e1, e2


MODES = ("single_launch", "workspace", "two_calls")


@pytest.mark.parametrize("mode", MODES)
def test_multiswag_grid_vs_reference(mode, ops, orc, swag_states, inputs):
    z = load_golden("case_multiswag_grid.npz")
    tp = tape(z)
    x = inputs["slow"]
    seed_idx, z1, z2, e1, e2 = _grid_from_tape(tp)
    eps = np.stack([np.stack([a, b], axis=1) for a, b in zip(e1, e2)])
    wa, w2, pd = stack_states(swag_states, z["ensemble"])
    if mode != "two_calls":
        out = ops.multiswag(dev(x), dev(wa), dev(w2), dev(pd), torch.as_tensor(seed_idx), dev(z1), dev(z2), dev(eps),
                            single_launch=(mode == "single_launch"))
    else:
        W = ops.swag_draw(dev(wa), dev(w2), dev(pd), torch.as_tensor(seed_idx), dev(z1), dev(z2))
        out = ops.forward(dev(x), W, eps=dev(eps))
    out = out.cpu().numpy()
    nbad, mx = close_report(out, z["out"])
    assert nbad == 0, (nbad, mx)
    o = orc.multiswag(x, wa, w2, pd, seed_idx, z1, z2, eps, sched=kernel_schedule(ops, orc))
    assert np.abs(out - o).max() <= 2e-6


@pytest.mark.parametrize("mode", MODES)
def test_chunk_loop_vs_reference(mode, ops, orc, swag_states, inputs):
    """figures/multiswag_5_planet.py:295-298: samples x torch.chunk(X, 10), one (seed, draw) per chunk per sample."""
    z = load_golden("case_chunk_loop.npz")
    tp = tape(z)
    B, nch, S = int(z["nrows"]), int(z["chunks"]), int(z["samples"])
    x = inputs["slow"][:B]
    seed_idx, z1, z2, e1, e2 = _grid_from_tape(tp)
    csz = -(-B // nch)
    eps = np.zeros((S, B, 2, 20), np.float32)
    for e in range(S * nch):
        s, c = divmod(e, nch)
        eps[s, c * csz:(c + 1) * csz, 0] = e1[e]
        eps[s, c * csz:(c + 1) * csz, 1] = e2[e]
    wa, w2, pd = stack_states(swag_states, z["ensemble"])
    if mode != "two_calls":
        out = ops.multiswag(dev(x), dev(wa), dev(w2), dev(pd), torch.as_tensor(seed_idx), dev(z1), dev(z2), dev(eps), nchunks=nch,
                            single_launch=(mode == "single_launch"))
    else:
        W = ops.swag_draw(dev(wa), dev(w2), dev(pd), torch.as_tensor(seed_idx), dev(z1), dev(z2))
        out = ops.forward(dev(x), W, eps=dev(eps), nchunks=nch)
    out = out.cpu().numpy()
    assert out.shape == z["out"].shape
    nbad, mx = close_report(out, z["out"])
    assert nbad == 0, (nbad, mx)


def _synthetic(B, seed=7):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(B, 1, 41, generator=g)
    x = base + 0.1 * torch.randn(B, 100, 41, generator=g)
    x[:, :, 0] = torch.linspace(-1.71, 1.74, 100)[None]
    return x.float().contiguous()


@pytest.mark.parametrize("B", (1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 200, 333))
def test_ragged_sizes_match_oracle(B, ops, orc, swag_states):
    """Every tail shape of the 4-system / 16-system / 64-system tiling, dense grid with 3 draws."""
    x = _synthetic(B).numpy()
    rng = np.random.default_rng(B)
    wa, w2, pd = stack_states(swag_states, (0, 12))
    J = 3
    seed_idx = np.array([0, 1, 1], np.int32)
    z1 = rng.standard_normal((J, 7583), dtype=np.float32)
    z2 = rng.standard_normal((J, 30), dtype=np.float32)
    eps = rng.standard_normal((J, B, 2, 20), dtype=np.float32)
    args = (dev(x), dev(wa), dev(w2), dev(pd), torch.as_tensor(seed_idx), dev(z1), dev(z2), dev(eps))
    out = ops.multiswag(*args, single_launch=True).cpu().numpy()
    o = orc.multiswag(x, wa, w2, pd, seed_idx, z1, z2, eps, sched=kernel_schedule(ops, orc))
    assert np.isfinite(out).all()
    assert np.abs(out - o).max() <= 2e-6, np.abs(out - o).max()
    # the draw-once workspace path and a forced 64-system block size give the same bits
    assert np.array_equal(ops.multiswag(*args, single_launch=False).cpu().numpy(), out)
    assert np.array_equal(ops.multiswag(*args, single_launch=True, systems_per_block=64).cpu().numpy(), out)


def test_philox_mode_is_the_explicit_mode_on_generated_noise(ops, swag_states):
    """In-kernel Philox noise == explicit noise filled by bnn_philox_normal_f32, bit for bit; and it is sharding invariant."""
    B, J, seed = 70, 4, 1234
    x = dev(_synthetic(B).numpy())
    wa, w2, pd = (dev(a) for a in stack_states(swag_states, (0, 12)))
    seed_idx = torch.tensor([0, 1, 0, 1], dtype=torch.int32)
    a = ops.multiswag(x, wa, w2, pd, seed_idx, philox_seed=seed, draw_id0=8, system_id0=100, single_launch=True)
    assert torch.equal(a, ops.multiswag(x, wa, w2, pd, seed_idx, philox_seed=seed, draw_id0=8, system_id0=100, single_launch=False))
    z1 = ops.philox_normal(0, seed, 8, J, width=7583)
    z2 = ops.philox_normal(1, seed, 8, J, width=30)
    eps = ops.philox_normal(2, seed, 8, J, B=B, system_id0=100)
    b = ops.multiswag(x, wa, w2, pd, seed_idx, z1, z2, eps)
    assert torch.equal(a, b)
    # sharded: second half of the systems, same global ids
    c = ops.multiswag(x[35:].contiguous(), wa, w2, pd, seed_idx, philox_seed=seed, draw_id0=8, system_id0=135)
    assert torch.equal(a[:, 35:], c)
    # the normals look normal
    n = eps.flatten().double().cpu().numpy()
    assert abs(n.mean()) < 0.02 and abs(n.std() - 1) < 0.02
    n = z1.flatten().double().cpu().numpy()
    assert abs(n.mean()) < 0.02 and abs(n.std() - 1) < 0.02 and abs((n ** 4).mean() - 3) < 0.15


def test_moments(ops):
    s = torch.rand(5, 37, 2, device="cuda")
    m = ops.moments(s)
    ref_m = torch.stack([s[..., 0].double().sum(0), (s[..., 0].double() ** 2).sum(0), s[..., 1].double().sum(0),
                         (s[..., 1].double() ** 2).sum(0)], 1)
    assert torch.allclose(m, ref_m, rtol=1e-12)
    m2 = ops.moments(s, m.clone())
    assert torch.allclose(m2, 2 * ref_m, rtol=1e-12)


def test_error_codes(ops):
    from bnn_chaos_model_amd import _native as N
    x = torch.zeros(2, 100, 40, device="cuda")
    with pytest.raises(NotImplementedError):
        ops.forward(x, torch.zeros(1, 7583, device="cuda"))
    with pytest.raises(N.NativeError):  # T = 1
        ops.forward(torch.zeros(2, 1, 41, device="cuda"), torch.zeros(1, 7583, device="cuda"))
    assert ops.forward(torch.zeros(2, 99, 41, device="cuda"), torch.zeros(1, 7583, device="cuda")).shape == (1, 2, 2)   # any T >= 2
    with pytest.raises(N.NativeError):
        N.Plan(0, hidden=129)


def test_all_thirty_pretrained_seeds(ops, orc):
    """The real 30-member ensemble in ONE launch (seed_idx = 0..29, the reference's taped normals per member) against the
    reference's own forward_swag_fast outputs (1e-5 relative, no exceedances), its sampled weights, and the oracle bit for bit
    -- including the five members with a negative variance element (v50_3, 12, 22, 25, 26)."""
    ens = load_golden("ensemble_v50.npz")
    z = load_golden("case_all_seeds.npz")
    wa, w2, pd = dev(ens["w_avg"]), dev(ens["w2_avg"]), dev(ens["pre_D"])
    idx = torch.arange(30, dtype=torch.int32)
    W = ops.swag_draw(wa, w2, pd, idx, dev(z["z1"]), dev(z["z2"]), scale=0.5).cpu().numpy()
    assert np.abs(W.astype(np.float64) - z["w"]).max() <= 2e-6
    for single in (False, True):
        out = ops.multiswag(dev(z["x"]), wa, w2, pd, idx, dev(z["z1"]), dev(z["z2"]), dev(z["eps"]), single_launch=single).cpu().numpy()
        nbad, mx = close_report(out, z["out"])
        assert nbad == 0, (single, nbad, mx)
    sched = kernel_schedule(ops, orc)
    for i in (3, 12, 22, 25, 26, 0, 29):
        w_orc = orc.swag_draw(ens["w_avg"][i], ens["w2_avg"][i], ens["pre_D"][i], z["z1"][i], z["z2"][i], scale=0.5)
        assert np.array_equal(W[i], w_orc), i
        o_k = orc.forward(z["x"], w_orc, z["eps"][i, :, 0], z["eps"][i, :, 1], sched=sched)
        assert np.abs(out[i] - o_k).max() <= 2e-6, i
