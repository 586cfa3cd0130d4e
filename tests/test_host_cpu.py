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
    declared = set(re.findall(r"^\s*(?:int|size_t|const char\*)\s+(bnn_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 14
    lib = C.CDLL(N.SO_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(N.EXPORTS)
    assert lib.bnn_abi_version() == N.ABI_VERSION == int(re.search(r"#define BNN_ABI_VERSION (\d+)", hdr).group(1))


def test_arch_validation_and_error_strings(N):
    L = N.lib()
    ok = N.BnnArch(41, 40, 20, 0, V50_MASK, 0.5, 0, 1, 1)
    assert L.bnn_param_count(C.byref(ok)) == 7583
    other = N.BnnArch(41, 32, 20, 0, V50_MASK, 0.5, 0, 1, 1)                     # another width: the generic engine's network
    assert L.bnn_param_count(C.byref(other)) == 41 + 40 + (32 * 41 + 32) + (32 * 32 + 32) + (20 * 32 + 20) + (32 * 40 + 32) + (32 * 32 + 32) + (2 * 32 + 2)
    lin = N.BnnArch(82, 40, 8, 1, V50_MASK, 0.5, 0, 0, 0)                         # in = out = 0: one Linear each; 82 features; fix_megno
    assert L.bnn_param_count(C.byref(lin)) == 82 + 18 + (8 * 82 + 8) + (2 * 18 + 2)
    for bad in (N.BnnArch(41, 129, 20, 0, V50_MASK, 0.5, 0, 1, 1), N.BnnArch(41, 40, 65, 0, V50_MASK, 0.5, 0, 1, 1),
                N.BnnArch(40, 40, 20, 0, V50_MASK, 0.5, 0, 1, 1), N.BnnArch(41, 40, 20, 0, V50_MASK, 0.5, 0, 8, 8),
                N.BnnArch(41, 40, 20, 0, V50_MASK, 0.5, 0, -1, 1), N.BnnArch(41, 128, 20, 0, V50_MASK, 0.5, 0, 3, 1)):
        assert L.bnn_param_count(C.byref(bad)) == N.ERR_UNSUPPORTED
        assert len(L.bnn_last_error()) > 20
    assert L.bnn_param_count(None) == N.ERR_INVALID
    bad = N.BnnArch(41, 40, 20, 0, 1 << 45, 0.5, 0, 1, 1)
    assert L.bnn_param_count(C.byref(bad)) == N.ERR_INVALID


def _order(N, mask, layer, noisy):
    buf = np.zeros(64, np.int32)
    a = N.BnnArch(41, 40, 20, 0, mask, 0.5, 0, 1, 1)
    n = N.check(N.lib().bnn_layer_order(C.byref(a), layer, noisy, buf.ctypes.data, 64))
    return buf[:n].copy()


def _table(N, mask, noisy, which):
    a = N.BnnArch(41, 40, 20, 0, mask, 0.5, 0, 1, 1)
    n = N.check(N.lib().bnn_fragment_table(C.byref(a), noisy, which, None, 0))
    t = np.zeros(n, np.int16)
    N.check(N.lib().bnn_fragment_table(C.byref(a), noisy, which, t.ctypes.data, n))
    return t.reshape(-1, 64)


def _image(N, mask, noisy):
    a = N.BnnArch(41, 40, 20, 0, mask, 0.5, 0, 1, 1)
    n = N.check(N.lib().bnn_fragment_table(C.byref(a), noisy, 1, None, 0))
    t = np.zeros(n, np.int16)
    N.check(N.lib().bnn_fragment_table(C.byref(a), noisy, 1, t.ctypes.data, n))
    return t.astype(np.int64)


def _wr(kin):
    """Weight-register counts of the feature_nn 4x4x1 path (bnn_layout.h, WR<KIN>): MFMAs per layer / 16, rounded up."""
    r1, r2, r3 = (kin * 10 + 15) // 16, (40 * 10 + 15) // 16, (40 * 5 + 15) // 16
    return dict(R1=r1, R2=r2, R3=r3, NR=r1 + r2 + r3)


OFF = dict(W1=81, B1=1721, W2=1761, B2=3361, W3=3401, B3=4201, W4=4221, B4=5821, W5=5861, B5=7461, W6=7501, B6=7581, D=7583)


@pytest.mark.parametrize("mask,noisy,kin", [(V50_MASK, 0, 31), (V50_MASK, 1, 41), (0, 0, 41), (1 << 7, 0, 41)])
def test_accumulation_orders_are_permutations(N, mask, noisy, kin):
    live = [c for c in range(41) if noisy or not (mask >> c) & 1]
    assert _order(N, mask, 0, noisy).tolist() == live           # bias first (the accumulator's start), then ascending inputs
    for layer in (1, 2):
        assert _order(N, mask, layer, noisy).tolist() == list(range(40))
    for layer in (3, 4, 5):
        assert sorted(_order(N, mask, layer, noisy).tolist()) == list(range(40))
    assert _image(N, mask, noisy).size == _wr(kin)["NR"] * 64


def test_operand_tables_use_every_parameter_exactly_once(N):
    """Each weight of feature_nn (register-resident A operands of the 4x4x1 path) / each weight and bias of regress_nn (16x16x4
    fragments) sits in exactly one slot; unused lanes read the zero slot."""
    img = _image(N, V50_MASK, 0)
    o = _wr(31)
    assert img.size == o["NR"] * 64 == 58 * 64
    body = img[img != OFF["D"]]
    live = [0] + list(range(8, 38))
    want = {OFF["W1"] + n * 41 + c for n in range(40) for c in live} | set(range(OFF["W2"], OFF["B2"])) | set(range(OFF["W3"], OFF["B3"]))
    assert len(body) == len(want) and set(body.tolist()) == want
    # only the last register of a layer has unused lanes: 310, 400, 200 MFMAs x 4 lanes
    per_reg = (img.reshape(-1, 64) != OFF["D"]).sum(1)
    assert per_reg[:o["R1"]].sum() == 1240 and per_reg[o["R1"]:o["R1"] + o["R2"]].sum() == 1600 and per_reg[o["R1"] + o["R2"]:].sum() == 800
    assert (per_reg[:o["R1"] - 1] == 64).all() and (per_reg[o["R1"]:o["R1"] + o["R2"]] == 64).all()
    f2 = _table(N, V50_MASK, 0, 2)
    a2 = f2[:70].ravel()
    a2 = a2[a2 != OFF["D"]]
    want2 = set(range(OFF["W4"], OFF["B4"])) | set(range(OFF["W5"], OFF["B5"])) | set(range(OFF["W6"], OFF["B6"]))
    assert len(a2) == len(want2) and set(a2.tolist()) == want2
    # any other mask: all 41 columns are multiplied, the masked ones read the zero slot
    g = _image(N, 1 << 7, 0)
    og = _wr(41)
    assert g.size == og["NR"] * 64 == 64 * 64
    l1 = g[:og["R1"] * 64]
    assert set(l1[l1 != OFF["D"]].tolist()) == {OFF["W1"] + n * 41 + c for n in range(40) for c in range(41) if c != 7}
    n1 = _image(N, V50_MASK, 1)[:og["R1"] * 64]                  # noisy forward: every column live (masked ones carry noise)
    assert set(n1[n1 != OFF["D"]].tolist()) == set(range(OFF["W1"], OFF["B1"]))


def test_register_dataflow_emulation_matches_oracle(N, inputs):
    """Evaluate feature_nn for a few rows in numpy exactly the way the kernel issues it (v_mfma_f32_4x4x1 with CBSZ = 4: lane = row,
    MFMA m = k * G + n of a layer takes its A operand -- neurons 4n..4n+3 against input k -- from lanes 4(m & 15).. of weight
    register m >> 4, accumulator register i of group n = neuron 4n + i; bias first, then the inputs in ascending order) and
    compare with the oracle's latents."""
    from oracle import oracle as orc
    z = load_golden("case_swagfast_v50_0_slow.npz")
    w = np.concatenate([z["w"], [0.0]]).astype(np.float64)
    regs = w[_image(N, V50_MASK, 0)].reshape(-1, 64)
    o = _wr(31)
    live = [0] + list(range(8, 38))
    x = inputs["slow"][0][:8].astype(np.float64)[:, live]     # 8 rows x 31 live columns

    def layer(xin, reg0, G, bias_off):
        out = np.tile(w[bias_off:bias_off + 4 * G], (xin.shape[0], 1))
        for m in range(xin.shape[1] * G):
            k, n, a = m // G, m % G, m & 15
            for i in range(4):
                out[:, 4 * n + i] += regs[reg0 + (m >> 4), 4 * a + i] * xin[:, k]
        return out

    h = np.maximum(layer(x, 0, 10, OFF["B1"]), 0)
    h2 = np.maximum(layer(h, o["R1"], 10, OFF["B2"]), 0)
    lat = layer(h2, o["R1"] + o["R2"], 5, OFF["B3"])
    tp1, tp2 = z["tape_002"], z["tape_003"]
    _, ex = orc.forward(inputs["slow"][:1], z["w"], tp1[:1], tp2[:1], extras=True)
    assert np.abs(lat - ex["latents"][0, :8]).max() < 5e-5


def test_fix_megno_layout_tables(N):
    """hparams['fix_megno'] (bnn_arch.fix_megno = 1): d = 7665, regress_nn.0 accumulates 42 inputs in 11 k-steps, every parameter of
    the wider layer sits in exactly one fragment slot, feature_nn's registers hold the same weights two floats further on."""
    L = N.lib()
    a = N.BnnArch(41, 40, 20, 1, V50_MASK, 0.5, 0, 1, 1)
    assert L.bnn_param_count(C.byref(a)) == 7665
    bad = N.BnnArch(41, 40, 20, 2, V50_MASK, 0.5, 0, 1, 1)
    assert L.bnn_param_count(C.byref(bad)) == N.ERR_INVALID

    def order(layer):
        buf = np.zeros(64, np.int32)
        n = N.check(L.bnn_layer_order(C.byref(a), layer, 0, buf.ctypes.data, 64))
        return buf[:n].tolist()

    def table(which):
        n = N.check(L.bnn_fragment_table(C.byref(a), 0, which, None, 0))
        t = np.zeros(n, np.int16)
        N.check(L.bnn_fragment_table(C.byref(a), 0, which, t.ctypes.data, n))
        return t.astype(np.int64)

    assert sorted(order(3)) == list(range(42)) and order(3)[-2:] == [40, 41]
    assert sorted(order(4)) == list(range(40)) and order(0) == [0] + list(range(8, 38))
    off_w4 = 41 + 42 + 1640 + 40 + 1600 + 40 + 800 + 20
    f2 = table(2).reshape(-1, 64)
    assert f2.shape[0] == 33 + 30 + 10 + 12 + 12 + 4
    a2 = f2[:33 + 30 + 10].ravel()
    a2 = a2[a2 != 7665]
    want = set(range(off_w4, off_w4 + 40 * 42)) | set(range(off_w4 + 1680 + 40, off_w4 + 1680 + 40 + 1600)) | \
        set(range(off_w4 + 1680 + 40 + 1640, off_w4 + 1680 + 40 + 1640 + 80))
    assert len(a2) == len(want) and set(a2.tolist()) == want
    w1 = table(1)
    w1 = w1[w1 != 7665]
    w1_plain = _image(N, V50_MASK, 0)
    w1_plain = w1_plain[w1_plain != OFF["D"]]
    assert np.array_equal(w1, w1_plain + 2)                      # everything behind summary_noise_logvar moves by two


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


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference (build container only)")
def test_save_swag_file_loads_in_the_unmodified_reference(tmp_path):
    """f3 from the reference's side: a checkpoint written by OUR save_swag (spock_reg_model.py:911-920), read by the REFERENCE's
    load_swag (:922-967), holds the same tensors / hparams / swa_params / K / c, and one reference forward_swag_fast on it reproduces
    the reference's own fixture bit for bit (tests/golden/check_save_swag_in_reference.py, run in a subprocess so that the reference
    module never enters this process)."""
    import subprocess
    import sys
    from bnn_chaos_model_amd import spock_reg_model as srm
    z = load_golden("swag_v50_12.npz")
    m = srm.SWAGModel(json.loads(str(z["hparams_json"]))).init_params(json.loads(str(z["swa_params_json"])))
    m.w_avg, m.w2_avg, m.pre_D = torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]), torch.tensor(z["pre_D"])
    path = str(tmp_path / "steps=300000_v50_12_output.pkl")
    srm.save_swag(m, path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "check_save_swag_in_reference.py"), path, "12"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["ok"] and all(rep["checks"].values()) and len(rep["checks"]) >= 10


def test_mask_helpers_are_the_references_subtractions():
    """zero_megno / zero_mmr / zero_nan / zero_eplusminus are `x - mask` (spock_reg_model.py:452-478): 0 for finite values, NaN for NaN and
    +-inf in the masked columns, everything else untouched; summarize_megno (:480-484) and set_flag (:410-414) exist with the reference's
    meaning.  Host-side torch, no GPU."""
    from bnn_chaos_model_amd import spock_reg_model as srm
    z = load_golden("swag_v50_0.npz")
    m = srm.SWAGModel(json.loads(str(z["hparams_json"]))).init_params(json.loads(str(z["swa_params_json"])))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 7, 41, generator=g)
    x[0, 1, 3] = float("inf"); x[1, 2, 38] = float("nan"); x[2, 3, 7] = float("-inf"); x[2, 4, 12] = float("inf")
    for fn, cols in ((m.zero_megno, [7]), (m.zero_mmr, [3, 6]), (m.zero_nan, [38, 39, 40]), (m.zero_eplusminus, [1, 2, 4, 5])):
        y = fn(x)
        other = [c for c in range(41) if c not in cols]
        assert torch.equal(torch.nan_to_num(y[..., other], nan=7.0, posinf=8.0, neginf=9.0), torch.nan_to_num(x[..., other], nan=7.0, posinf=8.0, neginf=9.0))
        want = x[..., cols] - x[..., cols]
        assert torch.equal(torch.isnan(y[..., cols]), torch.isnan(want)) and (y[..., cols][~torch.isnan(want)] == 0).all()
    allm = m._masked(x)
    assert torch.isnan(allm[0, 1, 3]) and torch.isnan(allm[1, 2, 38]) and torch.isnan(allm[2, 3, 7]) and torch.isinf(allm[2, 4, 12])
    sm = m.summarize_megno(x[:2])
    assert sm.shape == (2, 2) and torch.allclose(sm[:, 0], x[:2, :, 7].mean(1)) and torch.allclose(sm[:, 1], x[:2, :, 7].std(1))
    m.set_flag("random_sample", True)
    assert m.random_sample is True
    m.set_flag("random_sample", False)


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
    # any network the reference builds from hparams (spock_reg_model.py:301-321, 346-362) has the reference's state_dict layout
    m64 = srm.SWAGModel({**hp, "hidden": 64, "latent": 16, "in": 2, "out": 0, "include_derivatives": True})
    sd = m64.state_dict()
    assert [tuple(v.shape) for v in sd.values()] == [(82,), (32,), (64, 82), (64,), (64, 64), (64,), (64, 64), (64,), (16, 64), (16,), (2, 32), (2,)]
    assert list(sd.keys())[2:] == ["feature_nn.0.weight", "feature_nn.0.bias", "feature_nn.2.weight", "feature_nn.2.bias", "feature_nn.4.weight",
                                   "feature_nn.4.bias", "feature_nn.6.weight", "feature_nn.6.bias", "regress_nn.weight", "regress_nn.bias"]
    with pytest.raises(NotImplementedError):
        srm.SWAGModel({**hp, "hidden": 200})                             # widths above 128: no kernel


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


def test_integration_bindings_import_as_documented():
    """the star import, the sys.modules alias and the shim directory on sys.path all give a module with
    the reference's public names; `from spock import FeatureRegressor, FeatureRegressorXGB` (figures/multiswag_5_planet.py:28-29)
    resolves.  Run in a clean interpreter so that nothing imported earlier can mask a failure."""
    import subprocess
    code = r'''
import sys
ns = {}
exec("from bnn_chaos_model_amd.spock_reg_model import *", ns)
for name in ("load_swag", "save_swag", "VarModel", "SWAGModel", "soft_clamp", "EPSILON", "copy", "mlp"):
    assert name in ns, name
import bnn_chaos_model_amd.spock_reg_model as m
sys.modules["spock_reg_model"] = m
import spock_reg_model
assert spock_reg_model.load_swag is m.load_swag
del sys.modules["spock_reg_model"]
import os, bnn_chaos_model_amd
sys.path.insert(0, os.path.join(os.path.dirname(bnn_chaos_model_amd.__file__), "shims"))
import spock_reg_model as s2
assert s2.load_swag is m.load_swag and s2.SWAGModel is m.SWAGModel and s2.copy is m.copy
import spock
from spock import FeatureRegressor, FeatureRegressorXGB
from bnn_chaos_model_amd.regression import FeatureRegressor as FR
assert FeatureRegressor is FR
try:
    FeatureRegressorXGB()
except NotImplementedError:
    pass
else:
    raise AssertionError("the XGBoost baseline is out of scope and must say so")
print("ok")
'''
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a plain invocation starts two ranks itself (torch.distributed.run child, 127.0.0.1
    rendezvous); --launcher-selftest makes the ranks stop after the rendezvous + an all-reduce of ones (no GPU here)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2
    # under a launcher the rank count must agree with --gpus
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], cwd=ROOT, env=env2,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stdout + r.stderr)


def _gloo_bands_worker(rank, world, port, n_sims, trios, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from bnn_chaos_model_amd.distributed import all_gather_moments, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = n_sims * trios
        lo, hi = shard_bounds(B, world, trios)[rank]
        assert lo % trios == 0 and hi % trios == 0          # a simulation's trios stay on one rank
        sims = torch.arange(lo // trios, hi // trios, dtype=torch.float32)
        local = torch.stack([sims * 10 + k for k in range(6)], 1)   # stands in for [percentiles..., mean] of this rank's simulations
        out = all_gather_moments(local, n_sims)
        if rank == 0:
            q.put(out.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_sims,world", ((10, 2), (7, 2), (5, 3), (2, 3)))
def test_sharded_bands_gather_gloo(n_sims, world):
    """The gather of MultiSwagSharded.predictive_quantiles on CPU ranks: simulations (groups of 3 trios) are sharded whole,
    equal and ragged shards (and an empty one) reassemble in order with ONE all-gather."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + 10 * n_sims + world
    procs = [ctx.Process(target=_gloo_bands_worker, args=(r, world, port, n_sims, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([np.arange(n_sims, dtype=np.float32) * 10 + k for k in range(6)], 1)
    assert np.array_equal(got, want)


def test_headline_tile_loop_instruction_mix(N):
    """The headline kernel's time is the SUM of its matrix and vector instructions (nothing co-issues with the 4x4x1 MFMA), so the tile
    loop's vector-instruction count is its performance: 910 MFMAs + 121 other vector instructions per 64-row tile (80 ReLU, 20 + 20 packed
    Welford, 1).  Round 6 caught a refactoring that left every result bit-identical and cost 2.4 % -- the pool called through a lambda, its
    20 packed subtractions split into 20 scalar + 10 packed ones -- only in SQ_INSTS_VALU.  scripts/loop_histogram.py reads the loop out of
    the BUILT library: this test fails the CPU suite when the count moves."""
    import re
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "loop_histogram.py"),
                          "_ZN3bnn18bnn_forward_kernelILi31ELb0ELb0ELb0ELb0ELb0ELb0EEE", N.SO_PATH], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    counts = {m.group(2).strip(): int(m.group(1)) for m in re.finditer(r"^\s+(\d+)\s+(.+)$", out.stdout, re.M)}
    mfma = sum(v for k, v in counts.items() if k.startswith("MFMA"))
    vector = sum(v for k, v in counts.items() if k.startswith(("packed", "v_max_i32", "other vector", "transcendental", "v_mad_u64", "v_bitop3")))
    assert mfma == 910, out.stdout
    assert vector <= 121, out.stdout


def test_headline_kernels_use_no_scratch(N):
    """The register-resident forward forms sit at 236-256 VGPRs under __launch_bounds__(256, 2): a compiler update or a small edit would
    tip them into scratch silently.  Read the built library's code-object notes (scripts/resusage.py): the quiet and noisy forms of the
    pretrained network (31 / 41 columns, workspace and in-prologue draw, statistics tail) and the reduced-precision forms must use
    0 bytes of scratch and spill no VGPR; the fix_megno noisy form is known to spill 3 and is only bounded."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import resusage
    ks = resusage.kernels(N.SO_PATH)
    fwd = [k for k in ks if k["name"].startswith("bnn::bnn_forward_kernel<")]
    assert len(fwd) >= 15    # (incl. the two tile-split forms of the small grids, bnn_fwd_small.hip)
    for k in fwd:
        args = [a.strip() for a in k["name"].split("<", 1)[1].rstrip(">").split(",")]   # KIN, FUSED, NOISY, STATS, MEGNO, XNOISE
        megno_noisy = args[2] == "true" and args[4] == "true"
        assert k["vgpr"] <= 256 and k["agpr"] == 0, k
        if megno_noisy:
            assert k["scratch"] <= 32, k
        else:
            assert k["scratch"] == 0 and k["vgpr_spills"] == 0, k
    lowp = [k for k in ks if "bnn_forward_lowp_kernel" in k["name"]]
    assert len(lowp) == 5 and all(k["scratch"] == 0 and k["vgpr_spills"] == 0 for k in lowp)
    gen = {k["name"]: k for k in ks if "bnn_forward_generic_kernel" in k["name"]}
    assert len(gen) == 10
    # the generic engine: no scratch in the 41-feature one-wave-per-SIMD forms; the eight-wave (256-register) forms spill a bounded amount
    for name, k in gen.items():
        if name.endswith("true>"):
            assert k["vgpr"] <= 256 and k["agpr"] == 0 and k["scratch"] <= 320, k
        elif name.startswith("bnn::bnn_forward_generic_kernel<11"):
            assert k["scratch"] == 0, k
    # the pretrained network's specialised forms compiled into the library: eight waves (256 registers), no scratch, LDS without pool rows
    emb = {k["name"]: k for k in ks if k["name"].startswith("bnn_spec_forward_v50")}
    assert sorted(emb) == ["bnn_spec_forward_v50n", "bnn_spec_forward_v50q"]
    for k in emb.values():
        assert k["vgpr"] <= 256 and k["agpr"] == 0 and k["scratch"] == 0 and k["vgpr_spills"] == 0 and 0 < k["lds"] <= 80 * 1024, k
    # the non-finite scan streams x at the copy rate: few registers (many waves in flight), no scratch; the exact re-evaluation likewise
    nf = {k["name"]: k for k in ks if "bnn_nonfinite_" in k["name"]}
    assert sorted(n.split("::")[-1] for n in nf) == ["bnn_nonfinite_fixup_kernel<false>", "bnn_nonfinite_fixup_kernel<true>", "bnn_nonfinite_reset_kernel",
                                                     "bnn_nonfinite_scan_kernel", "bnn_nonfinite_scan_small_kernel"]
    assert all(k["scratch"] == 0 and k["vgpr_spills"] == 0 for k in nf.values())
    assert [k for n, k in nf.items() if n.endswith("scan_kernel")][0]["vgpr"] <= 64
