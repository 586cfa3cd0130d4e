"""CPU-side checks: the C ABI library loads and exports what include/bnn_chaos_hip.h declares, host-side operand
tables, the checkpoint reader, the reference-surface constructor, and the N>1 sharding path on gloo.
No GPU compute is issued here."""
import ctypes as C
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, load_golden

V50_MASK = 0x1C0000000FE


@pytest.fixture(scope="module")
def N():
    from bnn_chaos_model_amd import _native
    _native.lib()
    return _native


def test_abi_exports_every_declared_symbol(N):
    hdr = open(os.path.join(ROOT, "include", "bnn_chaos_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(bnn_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 14
    lib = C.CDLL(N.SO_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(N.EXPORTS)
    assert lib.bnn_abi_version() == 1


def test_arch_validation_and_error_strings(N):
    L = N.lib()
    ok = N.BnnArch(41, 40, 20, 0, V50_MASK, 0.5, 0)
    assert L.bnn_param_count(C.byref(ok)) == 7583
    bad = N.BnnArch(41, 32, 20, 0, V50_MASK, 0.5, 0)
    assert L.bnn_param_count(C.byref(bad)) == N.ERR_UNSUPPORTED
    assert b"41->40->40->20" in L.bnn_last_error()
    assert L.bnn_param_count(None) == N.ERR_INVALID
    bad = N.BnnArch(41, 40, 20, 0, 1 << 45, 0.5, 0)
    assert L.bnn_param_count(C.byref(bad)) == N.ERR_INVALID


def _order(N, mask, layer, noisy):
    buf = np.zeros(64, np.int32)
    a = N.BnnArch(41, 40, 20, 0, mask, 0.5, 0)
    n = N.check(N.lib().bnn_layer_order(C.byref(a), layer, noisy, buf.ctypes.data, 64))
    return buf[:n].copy()


def _table(N, mask, noisy, which):
    a = N.BnnArch(41, 40, 20, 0, mask, 0.5, 0)
    n = N.check(N.lib().bnn_fragment_table(C.byref(a), noisy, which, None, 0))
    t = np.zeros(n, np.int16)
    N.check(N.lib().bnn_fragment_table(C.byref(a), noisy, which, t.ctypes.data, n))
    return t.reshape(-1, 64)


@pytest.mark.parametrize("mask,noisy,nk1", [(V50_MASK, 0, 8), (V50_MASK, 1, 11), (0, 0, 11), (1 << 7, 0, 11)])
def test_accumulation_orders_are_permutations(N, mask, noisy, nk1):
    live = [c for c in range(41) if noisy or not (mask >> c) & 1]
    o0 = _order(N, mask, 0, noisy)
    assert sorted(o0.tolist()) == sorted(live + [-1])            # every live column once + the bias slot
    for layer in (1, 2, 3, 4, 5):
        assert sorted(_order(N, mask, layer, noisy).tolist()) == list(range(40))
    assert _table(N, mask, noisy, 1).shape[0] == 3 * nk1 + 30 + 20 + 12 + 8


def test_fragment_tables_use_every_parameter_exactly_once(N):
    """A operands: each weight of feature_nn / regress_nn appears in exactly one (register, lane) slot."""
    OFF = dict(W1=81, B1=1721, W2=1761, B2=3361, W3=3401, B3=4201, W4=4221, B4=5821, W5=5861, B5=7461, W6=7501, B6=7581, D=7583)
    f1 = _table(N, V50_MASK, 0, 1)
    a1 = f1[:24 + 30 + 20].ravel()
    a1 = a1[a1 != OFF["D"]]
    live = [0] + list(range(8, 38))
    want = {OFF["W1"] + n * 41 + c for n in range(40) for c in live} | set(range(OFF["B1"], OFF["B1"] + 40)) | \
        set(range(OFF["W2"], OFF["B2"])) | set(range(OFF["W3"], OFF["B3"]))
    assert len(a1) == len(want) and set(a1.tolist()) == want
    b = f1[74:].ravel()
    b = b[b != OFF["D"]]
    assert set(b.tolist()) == set(range(OFF["B2"], OFF["B2"] + 40)) | set(range(OFF["B3"], OFF["B3"] + 20))
    f2 = _table(N, V50_MASK, 0, 2)
    a2 = f2[:70].ravel()
    a2 = a2[a2 != OFF["D"]]
    want2 = set(range(OFF["W4"], OFF["B4"])) | set(range(OFF["W5"], OFF["B5"])) | set(range(OFF["W6"], OFF["B6"]))
    assert len(a2) == len(want2) and set(a2.tolist()) == want2
    # generic (no mask) variant covers all 41 columns
    g1 = _table(N, 0, 0, 1)[:33].ravel()
    g1 = g1[g1 != OFF["D"]]
    assert set(g1.tolist()) == set(range(OFF["W1"], OFF["B1"] + 40))


def test_mfma_dataflow_emulation_matches_oracle(N, inputs):
    """Evaluate feature_nn for one 16-row tile in numpy exactly the way the kernel wires v_mfma_f32_16x16x4_f32
    (A = table-gathered weights, B = activations, accumulators feed the next layer) and compare with the oracle."""
    from oracle import oracle as orc
    z = load_golden("case_swagfast_v50_0_slow.npz")
    w = np.concatenate([z["w"], [0.0]]).astype(np.float32)
    f1 = _table(N, V50_MASK, 0, 1).astype(np.int64)
    x = inputs["slow"][0][:16]  # 16 rows = lane columns c
    lane = np.arange(64)
    g, c = lane >> 4, lane & 15

    def mfma(a, b, acc):  # acc[i][lane] : C[4g+i][c]; A lane (g,m): A[m][k=g]; B lane (g,c): B[k=g][c]
        A = np.zeros((16, 4), np.float64); Bm = np.zeros((4, 16), np.float64)
        A[lane & 15, lane >> 4] = a
        Bm[lane >> 4, lane & 15] = b
        Cm = A @ Bm
        out = acc.copy()
        for i in range(4):
            out[i] += Cm[4 * g + i, c]
        return out

    def kmap_input(s, gg):
        if gg < 3:
            return 8 + 8 * gg + s
        return 32 + s if s < 6 else (0 if s == 6 else -1)

    b_in = np.zeros((8, 64))
    for s in range(8):
        for l in range(64):
            col = kmap_input(s, l >> 4)
            b_in[s, l] = 1.0 if col < 0 else x[l & 15, col]
    h = [np.zeros((4, 64)) for _ in range(3)]
    for s in range(8):
        for mt in range(3):
            h[mt] = mfma(w[f1[s * 3 + mt]], b_in[s], h[mt])
    h = [np.maximum(v, 0) for v in h]
    h2 = [np.stack([w[f1[74 + mt * 4 + i]] for i in range(4)]).astype(np.float64) for mt in range(3)]
    for ks in range(10):
        for mt in range(3):
            h2[mt] = mfma(w[f1[24 + ks * 3 + mt]], h[ks >> 2][ks & 3], h2[mt])
    h2 = [np.maximum(v, 0) for v in h2]
    y = [np.stack([w[f1[86 + mt * 4 + i]] for i in range(4)]).astype(np.float64) for mt in range(2)]
    for ks in range(10):
        for mt in range(2):
            y[mt] = mfma(w[f1[54 + ks * 2 + mt]], h2[ks >> 2][ks & 3], y[mt])
    lat = np.zeros((16, 20))
    for l in range(64):
        for i in range(4):
            lat[l & 15, 4 * (l >> 4) + i] = y[0][i, l]
        lat[l & 15, 16 + (l >> 4)] = y[1][0, l]
    tp1, tp2 = z["tape_002"], z["tape_003"]
    _, ex = orc.forward(inputs["slow"][:1], z["w"], tp1[:1], tp2[:1], extras=True)
    assert np.abs(lat - ex["latents"][0, :16]).max() < 5e-5


def test_checkpoint_roundtrip_and_reference_file(tmp_path):
    from bnn_chaos_model_amd import checkpoint, spock_reg_model as srm
    z = load_golden("swag_v50_0.npz")
    hp = json.loads(str(z["hparams_json"]))
    swa = json.loads(str(z["swa_params_json"]))
    p = tmp_path / "steps=1_v50_0_output.pkl"
    checkpoint.write_swag_file(str(p), hp, swa, torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"]))
    m = srm.load_swag(str(p))
    assert m.K == 30 and m.c == 5 and m.hparams.hidden == 40 and m.hparams["latent"] == 20
    assert torch.equal(m.w_avg, torch.tensor(z["w_avg"])) and torch.equal(m.pre_D, torch.tensor(z["pre_D"]))
    assert np.array_equal(m.ssX.mean_, z["ssX_mean"]) and np.array_equal(m.ssX.scale_, z["ssX_scale"])
    assert m.zero_mask() == V50_MASK and m.megno_location == 7
    srm.save_swag(m, str(tmp_path / "again_v50.pkl"))
    m2 = srm.load_swag(str(tmp_path / "again_v50.pkl"))
    assert torch.equal(m2.w2_avg, m.w2_avg)
    # a pickle that references anything outside the whitelist is refused
    evil = tmp_path / "evil.pkl"
    torch.save({"hparams": {}, "swa_params": {}, "w_avg": torch.zeros(1), "w2_avg": torch.zeros(1), "pre_D": torch.zeros(1, 1),
                "x": np.float64(1.0)}, str(evil))
    import pickle
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.read_swag_file(str(evil))
    ref = "/root/reference/pretrained"
    if os.path.isdir(ref):  # build container only: the real file, AttributeDict and all
        import glob
        f = glob.glob(ref + "/*_v50_0_output.pkl")[0]
        it = checkpoint.read_swag_file(f)
        assert np.array_equal(it["w_avg"].numpy(), z["w_avg"]) and np.array_equal(it["pre_D"].numpy(), z["pre_D"])
        assert it["hparams"].hidden == 40


def test_constructor_reproduces_reference_side_effects(tmp_path):
    """load_swag -> SWAGModel(hparams): seed_everything(seed) + the reference's module init order (spock_reg_model.py:343-362)."""
    from bnn_chaos_model_amd import checkpoint, spock_reg_model as srm
    z = load_golden("case_init_v50_3.npz")
    hp = json.loads(str(z["hparams_json"]))
    swa = json.loads(str(z["swa_params_json"]))
    p = tmp_path / "x_v50_3_output.pkl"
    d = 7583
    checkpoint.write_swag_file(str(p), hp, swa, torch.zeros(d), torch.zeros(d), torch.zeros(d, 30))
    m = srm.load_swag(str(p))
    assert np.array_equal(m.flatten().numpy(), z["init_flat"])          # random init identical to the reference's
    assert np.array_equal(torch.randn(4).numpy(), z["next_torch"])      # and the global generators are where it leaves them
    assert np.array_equal(np.random.rand(4), z["next_numpy"])
    sd = m.state_dict()
    assert list(sd.keys())[:3] == ["input_noise_logvar", "summary_noise_logvar", "feature_nn.0.weight"]
    v = torch.arange(d, dtype=torch.float32)
    m.load(v)
    assert torch.equal(m.flatten(), v) and torch.equal(m.state_dict()["regress_nn.4.bias"], v[-2:])
    with pytest.raises(NotImplementedError):
        srm.SWAGModel({**hp, "hidden": 64})


def test_standard_scaler_matches_reference_constants():
    from bnn_chaos_model_amd import spock_reg_model as srm
    z = load_golden("inputs.npz")
    ss = srm.v50_scaler()
    raw4 = np.ones((4, 100, 41)) * 4
    x = torch.tensor(ss.transform(raw4.reshape(-1, 41)).reshape(raw4.shape)).float().numpy()
    assert np.array_equal(x, z["x_const4"])  # the constant-4 "unstable" fill of figures/multiswag_5_planet.py:214-215


def test_shard_bounds():
    from bnn_chaos_model_amd.distributed import shard_bounds
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(8, 8) == [(i, i + 1) for i in range(8)]
    assert shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]


def _gloo_worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bnn_chaos_model_amd.distributed import sharded_predictive_moments
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = [np.load(os.path.join(GOLDEN, f"swag_v50_{i}.npz")) for i in (0, 12)]
        wa = np.stack([s["w_avg"] for s in st]); w2 = np.stack([s["w2_avg"] for s in st]); pd = np.stack([s["pre_D"] for s in st])
        x = np.load(os.path.join(GOLDEN, "inputs.npz"))["x_slow"][:B]
        rng = np.random.default_rng(5)  # identical draw list and noise on every rank (replicated, keyed by global ids)
        J = 3
        seed_idx = np.array([0, 1, 0], np.int32)
        z1 = rng.standard_normal((J, 7583), dtype=np.float32); z2 = rng.standard_normal((J, 30), dtype=np.float32)
        eps = rng.standard_normal((J, B, 2, 20), dtype=np.float32)

        def local(lo, hi):  # the oracle stands in for the GPU kernels: this test is about the sharding + gather
            if hi == lo:
                return torch.zeros((0, 4), dtype=torch.float64)
            s = orc.multiswag(x[lo:hi], wa, w2, pd, seed_idx, z1, z2, eps[:, lo:hi]).astype(np.float64)
            return torch.tensor(np.stack([s[..., 0].sum(0), (s[..., 0] ** 2).sum(0), s[..., 1].sum(0), (s[..., 1] ** 2).sum(0)], 1))

        mom = sharded_predictive_moments(local, B)
        if rank == 0:
            q.put(mom.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", (8, 7))
def test_sharded_moments_gloo_world2(B):
    """N>1 path on CPU: 2 ranks, systems sharded, one all-gather of moments == the single-process result."""
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + B
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    st = [np.load(os.path.join(GOLDEN, f"swag_v50_{i}.npz")) for i in (0, 12)]
    wa = np.stack([s["w_avg"] for s in st]); w2 = np.stack([s["w2_avg"] for s in st]); pd = np.stack([s["pre_D"] for s in st])
    x = np.load(os.path.join(GOLDEN, "inputs.npz"))["x_slow"][:B]
    rng = np.random.default_rng(5)
    z1 = rng.standard_normal((3, 7583), dtype=np.float32); z2 = rng.standard_normal((3, 30), dtype=np.float32)
    eps = rng.standard_normal((3, B, 2, 20), dtype=np.float32)
    s = orc.multiswag(x, wa, w2, pd, np.array([0, 1, 0], np.int32), z1, z2, eps).astype(np.float64)
    want = np.stack([s[..., 0].sum(0), (s[..., 0] ** 2).sum(0), s[..., 1].sum(0), (s[..., 1] ** 2).sum(0)], 1)
    assert got.shape == (B, 4) and np.array_equal(got, want)
