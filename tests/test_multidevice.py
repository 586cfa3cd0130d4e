"""One process, several GPUs (bnn_chaos_model_amd/multidevice.py): the partition / exchange logic on the CPU, and -- on the one-GPU box --
several logical shards on the same card, which must reproduce the single-shard result bit for bit (global Philox ids, chunks of the
whole batch).  Reference call sites: figures/multiswag_5_planet.py:61, 280-298 (a single-process script)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden


def _cpu_set(n):
    from bnn_chaos_model_amd.multidevice import DeviceSet
    ds = object.__new__(DeviceSet)
    ds.devices, ds._replicas, ds.last_exchange, ds.exchange, ds._h2d = [torch.device("cpu")] * n, {}, None, "copies", None
    return ds


@pytest.mark.parametrize("B,n,group", [(10, 3, 1), (9, 3, 3), (2, 4, 1), (30, 8, 3), (7, 1, 1)])
def test_partition_and_exchange_on_cpu(B, n, group):
    ds = _cpu_set(n)
    bounds = ds.bounds(B, group)
    assert bounds[0][0] == 0 and bounds[-1][1] == B and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert all((hi - lo) % group == 0 for lo, hi in bounds)
    rows = torch.arange(B // group, dtype=torch.float64)[:, None] * torch.ones(1, 4, dtype=torch.float64)
    seen = []
    def fn(i, dev, lo, hi):
        seen.append((i, lo, hi))
        return rows[lo // group: hi // group].clone()
    parts = ds.run(B, fn, group=group)
    assert [p is None for p in parts] == [hi == lo for lo, hi in bounds]      # empty shards (more devices than simulations) launch nothing
    got = ds.gather_rows(parts)
    assert torch.equal(got, rows) and ds.last_exchange is not None
    st = ds.replicate("k", (torch.ones(3), torch.arange(4)))
    assert len(st) == n and all(torch.equal(s[0], torch.ones(3)) for s in st)
    assert ds.replicate("k", (torch.ones(3),)) is not st
    # staging: every shard's rows, in shard order, before anything runs (pageable host source: one host thread per shard)
    X = torch.arange(B * 6, dtype=torch.float64).reshape(B, 2, 3)
    xs = ds.stage(X, group=group)
    assert [x is None for x in xs] == [hi == lo for lo, hi in bounds]
    assert torch.equal(torch.cat([x for x in xs if x is not None]), X.float()) and all(x.dtype == torch.float32 for x in xs if x is not None)
    info = ds.h2d_ms()
    assert info["bytes"] == X.numel() * 8 and info["host_ms"] >= 0.0 and "pageable" in info["mode"]


def test_which_devices(monkeypatch):
    """None = the CURRENT device (a rank of a process-per-GPU launch, or a script that called set_device, keeps its GPU); "all" opts in to
    every visible GPU -- except under a launcher; BNN_CHAOS_DEVICES stands in for the argument of a script that cannot be edited."""
    from bnn_chaos_model_amd import multidevice as md
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 5)
    for k in ("WORLD_SIZE", "LOCAL_RANK", "BNN_CHAOS_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    dev = lambda *i: [torch.device("cuda", j) for j in i]
    assert md.resolve_devices(None) == dev(5)
    assert md.resolve_devices("all") == dev(*range(8))
    assert md.resolve_devices(3) == dev(0, 1, 2)
    assert md.resolve_devices([2, 2, "cuda:7"]) == dev(2, 2, 7)
    monkeypatch.setenv("BNN_CHAOS_DEVICES", "all")
    assert md.resolve_devices(None) == dev(*range(8)) and md.resolve_devices([1]) == dev(1)
    monkeypatch.setenv("BNN_CHAOS_DEVICES", "1,3")
    assert md.resolve_devices(None) == dev(1, 3)
    monkeypatch.setenv("LOCAL_RANK", "5")          # torch.distributed.run: every GPU visible to every rank
    monkeypatch.setenv("BNN_CHAOS_DEVICES", "all")
    assert md.resolve_devices(None) == dev(5) and md.resolve_devices("all") == dev(5)
    monkeypatch.delenv("LOCAL_RANK")
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert md.resolve_devices("all") == dev(5)
    with pytest.raises(ValueError):
        md.resolve_devices([9])
    with pytest.raises(ValueError):
        md.resolve_devices("some")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        md.resolve_devices(None)


def test_exchange_choice_and_what_counts_as_rccl_absent(monkeypatch):
    """Peer copies by default; the RCCL form is opt-in and only RcclUnavailable falls back to the copies -- any other failure raises."""
    from bnn_chaos_model_amd import multidevice as md
    ds = _cpu_set(2)
    rows = [torch.ones(2, 3), torch.zeros(1, 3)]
    assert ds.gather_rows(rows).shape == (3, 3) and ds.last_exchange == "peer copies"
    ds.exchange = "rccl"     # CPU tensors / repeated devices: not RCCL's business
    assert ds.gather_rows(rows).shape == (3, 3) and "more than once" in ds.last_exchange

    class FakeDev(torch.Tensor):
        pass
    calls = []
    def boom(parts):
        calls.append(len(parts))
        raise md.RcclUnavailable("not built in")
    monkeypatch.setattr(md, "_rccl_all_gather", boom)
    # distinct "devices": fake it through the device list check by patching set() semantics is overkill -- call the branch directly
    class P:   # minimal stand-in for two tensors on two GPUs
        def __init__(self, t, i):
            self.t, self.device, self.shape, self.is_cuda = t, f"cuda:{i}", t.shape, True
        def to(self, d, non_blocking=False):
            return self.t
    out = ds.gather_rows([P(rows[0], 0), P(rows[1], 1)])
    assert calls == [2] and out.shape == (3, 3) and "unavailable" in ds.last_exchange
    monkeypatch.setattr(md, "_rccl_all_gather", lambda parts: (_ for _ in ()).throw(ValueError("a real bug")))
    with pytest.raises(ValueError, match="a real bug"):
        ds.gather_rows([P(rows[0], 0), P(rows[1], 1)])


@pytest.fixture(scope="module")
def fr(tmp_path_factory):
    from bnn_chaos_model_amd import checkpoint
    from bnn_chaos_model_amd.regression import FeatureRegressor
    d = tmp_path_factory.mktemp("pretrained_md")
    for i in (0, 12):
        z = load_golden(f"swag_v50_{i}.npz")
        checkpoint.write_swag_file(str(d / f"steps=300000_v50_{i:02d}_output.pkl"), json.loads(str(z["hparams_json"])),
                                   json.loads(str(z["swa_params_json"])), torch.tensor(z["w_avg"]), torch.tensor(z["w2_avg"]),
                                   torch.tensor(z["pre_D"]))
    return FeatureRegressor(cuda=True, filebase=str(d / "*v50*output.pkl"), sort=True)


@pytest.mark.gpu
@pytest.mark.parametrize("rng", ("torch", "philox"))
def test_logical_shards_reproduce_the_single_shard_result(fr, rng, inputs):
    """devices=[0, 0, 0]: three shards of a 77-system batch (chunks = 10: ragged chunks that straddle the shard boundaries) on one
    card == one shard, bit for bit, with the reference's RNG order and with in-kernel Philox; likewise the default (all visible)."""
    X = torch.tensor(np.tile(inputs["slow"], (3, 1, 1))[:77])
    outs = {}
    for name, devs in (("one", [0]), ("three", [0, 0, 0]), ("seven", [0] * 7), ("default", None), ("all", "all")):
        np.random.seed(5); torch.manual_seed(5)
        outs[name] = fr.sample_full_swag_many(X, samples=4, chunks=10, rng=rng, philox_seed=17, draw_id0=40, system_id0=1000, devices=devs)
        assert outs[name].shape == (4, 77, 2) and outs[name].device == X.device
    assert torch.equal(outs["one"], outs["three"]) and torch.equal(outs["one"], outs["seven"]) and torch.equal(outs["one"], outs["default"])
    assert torch.equal(outs["one"], outs["all"])
    info = fr.last_run["h2d"]()
    assert info["ms"] >= 0.0 and info["bytes"] == X.numel() * 4 and fr.last_run["devices"] == ["cuda:0"]
    # and the single-shard result is the loop of the reference's script (chunk by chunk through sample_full_swag), same seeds
    if rng == "torch":
        np.random.seed(5); torch.manual_seed(5)
        loop = torch.cat([torch.cat([fr.sample_full_swag(Xp) for Xp in torch.chunk(X, 10)])[None] for _ in range(4)])
        assert torch.equal(loop, outs["one"])


@pytest.mark.gpu
def test_bands_and_moments_over_logical_shards(fr, inputs):
    from bnn_chaos_model_amd import ops
    from bnn_chaos_model_amd.distributed import MultiSwagSharded
    X = torch.tensor(np.tile(inputs["slow"], (4, 1, 1))[:120])          # 40 simulations x 3 trios
    res = {}
    for name, devs in (("one", [0]), ("four", [0, 0, 0, 0]), ("many", [0] * 6)):
        np.random.seed(9)
        res[name] = fr.predictive_bands(X, samples=64, chunks=10, trios=3, philox_seed=3, system_id0=300, samples_per_launch=16, devices=devs)
    for name in ("four", "many"):
        assert torch.equal(res["one"]["percentiles"], res[name]["percentiles"]) and torch.equal(res["one"]["average"], res[name]["average"])
        assert "peer copies" in res[name]["exchange"]
    assert res["one"]["percentiles"].shape == (40, 5)
    # the dense grid through MultiSwagSharded(devices=...): moments and bands, against its own single-device form
    wa, w2, pd = fr.ensemble_state()
    seed_idx = torch.arange(24, dtype=torch.int32) % 2
    xg = X.cuda()
    single = MultiSwagSharded(wa, w2, pd, draws_per_launch=8)
    m1 = single.predictive_moments(xg, 120, seed_idx, philox_seed=4)
    q1 = single.predictive_quantiles(xg, 120, seed_idx, philox_seed=4, trios=3)
    multi = MultiSwagSharded(wa, w2, pd, draws_per_launch=8, devices=[0, 0, 0])
    assert torch.equal(m1, multi.predictive_moments(X, 120, seed_idx, philox_seed=4))       # x may live on the host
    assert torch.equal(q1, multi.predictive_quantiles(xg, 120, seed_idx, philox_seed=4, trios=3))
    assert MultiSwagSharded(wa, w2, pd, devices="all").predictive_moments(xg, 120, seed_idx, philox_seed=4).shape == (120, 4)


@pytest.mark.gpu
def test_another_network_through_the_batched_drivers(tmp_path):
    """An ensemble of two checkpoints of ANOTHER network (hidden 64, latent 16: the generic engine) through FeatureRegressor's batched
    drivers: the one-launch MC loop equals the script's chunk-by-chunk loop bit for bit (reference RNG order), logical shards on one
    card equal the single shard (chunks of the whole batch inside the generic kernel), and the streamed bands run."""
    from bnn_chaos_model_amd import checkpoint
    from bnn_chaos_model_amd.regression import FeatureRegressor
    z = load_golden("case_arch_h64l16.npz")
    hp = json.loads(str(z["hparams_json"]))
    for k, v in list(hp.items()):
        if isinstance(v, str) and v in ("True", "False"):
            hp[k] = v == "True"
    rng = np.random.default_rng(2)
    for i in range(2):
        jit = (1.0 + 0.01 * rng.standard_normal(z["w_avg"].shape)).astype(np.float32)
        checkpoint.write_swag_file(str(tmp_path / f"net64_{i}_output.pkl"), hp, json.loads(str(z["swa_params_json"])),
                                   torch.tensor(z["w_avg"] * jit), torch.tensor(z["w2_avg"] * jit * jit), torch.tensor(z["pre_D"] * jit[:, None]))
    fr = FeatureRegressor(cuda=True, filebase=str(tmp_path / "net64_*_output.pkl"), sort=True)
    assert len(fr.swag_ensemble) == 2 and fr.swag_ensemble[0].hparams["hidden"] == 64
    X = torch.tensor(np.tile(z["x"], (3, 1, 1))[:45])
    np.random.seed(4); torch.manual_seed(4)
    one = fr.sample_full_swag_many(X, samples=3, chunks=10, devices=[0])
    np.random.seed(4); torch.manual_seed(4)
    three = fr.sample_full_swag_many(X, samples=3, chunks=10, devices=[0, 0, 0])
    np.random.seed(4); torch.manual_seed(4)
    loop = torch.cat([torch.cat([fr.sample_full_swag(Xp) for Xp in torch.chunk(X, 10)])[None] for _ in range(3)])
    assert torch.equal(one, three) and torch.equal(one, loop)
    np.random.seed(5)
    a = fr.predictive_bands(X, samples=32, chunks=5, trios=3, philox_seed=8, samples_per_launch=8, devices=[0])
    np.random.seed(5)
    b = fr.predictive_bands(X, samples=32, chunks=5, trios=3, philox_seed=8, samples_per_launch=8, devices=[0, 0])
    assert torch.equal(a["percentiles"], b["percentiles"]) and a["percentiles"].shape == (15, 5)
    with pytest.raises(NotImplementedError):
        fr.sample_full_swag_many(X[:, :, :40], samples=1)
    # the same ensemble through its run-time-compiled kernels (FeatureRegressor.specialize): not a bit changes, on any shard
    assert fr.specialize() is fr and fr.swag_ensemble[0]._plan().spec_attached(False)
    np.random.seed(4); torch.manual_seed(4)
    assert torch.equal(one, fr.sample_full_swag_many(X, samples=3, chunks=10, devices=[0, 0, 0]))
    np.random.seed(5)
    c = fr.predictive_bands(X, samples=32, chunks=5, trios=3, philox_seed=8, samples_per_launch=8, devices=[0, 0])
    assert torch.equal(a["percentiles"], c["percentiles"])


@pytest.mark.gpu
def test_rccl_branch_executes_at_world_size_one():
    """multidevice._rccl_all_gather -- communicator creation inside the process, the padded all-gather, the trim -- on a one-device list:
    the code path a multi-GPU node takes with BNN_MULTIDEVICE_EXCHANGE=rccl, executed on the one card this box has.  (Between distinct
    GPUs it has not run from this environment: the exchange defaults to peer copies.)"""
    from bnn_chaos_model_amd import multidevice as md
    rows = torch.arange(28, dtype=torch.float64, device="cuda").reshape(7, 4)
    try:
        out = md._rccl_all_gather([rows])
    except md.RcclUnavailable as e:
        pytest.skip(f"torch.cuda.nccl is not available in this build: {e}")
    torch.cuda.synchronize()
    assert out.shape == (7, 4) and torch.equal(out, rows) and out.device == rows.device
    # float32 payload (the bands) and a second call on the cached communicator
    bands = torch.randn(5, 6, device="cuda")
    assert torch.equal(md._rccl_all_gather([bands]), bands)
    # through the DeviceSet with the RCCL exchange asked for: one shard needs no exchange; two shards on one card fall back, and say why
    ds = md.DeviceSet([0], exchange="rccl")
    assert torch.equal(ds.gather_rows([rows]), rows) and ds.last_exchange == "none (one shard)"
    ds2 = md.DeviceSet([0, 0], exchange="rccl")
    got = ds2.gather_rows([rows[:4], rows[4:]])
    assert torch.equal(got, rows) and "more than once" in ds2.last_exchange
    with pytest.raises(ValueError):
        md.DeviceSet([0], exchange="smoke signals")


@pytest.mark.gpu
def test_damaged_systems_through_the_sharded_drivers(fr, inputs):
    """Non-finite inputs through the batched drivers: every shard scans its own rows (a record per shard, the chunks still those of the
    whole batch), so logical shards reproduce the single-shard result -- NaN for the damaged systems / simulations, the clean ones bit
    for bit as on the clean batch."""
    X = torch.tensor(np.tile(inputs["slow"], (3, 1, 1))[:90].copy())
    Xb = X.clone()
    Xb[7, 3, 2] = float("nan")          # masked column: NaN in the reference (x - mask)
    Xb[33, 50, 12] = float("inf")       # live column
    Xb[89, 99, 40] = float("-inf")      # masked column, last row of the last shard
    hurt = [7, 33, 89]
    keep = [b for b in range(90) if b not in hurt]
    outs = {}
    for name, devs in (("one", [0]), ("four", [0, 0, 0, 0])):
        for rng in ("torch", "philox"):
            np.random.seed(5); torch.manual_seed(5)
            outs[name, rng] = fr.sample_full_swag_many(Xb, samples=3, chunks=10, rng=rng, philox_seed=17, devices=devs)
    for rng in ("torch", "philox"):
        a, b = outs["one", rng], outs["four", rng]
        assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))
        assert torch.isnan(a[:, hurt]).all() and torch.isfinite(a[:, keep]).all()
        np.random.seed(5); torch.manual_seed(5)
        clean = fr.sample_full_swag_many(X, samples=3, chunks=10, rng=rng, philox_seed=17, devices=[0], assume_finite=True)
        assert torch.equal(a[:, keep], clean[:, keep])
    # the script's own loop, call by call through the surface (default: assume_finite = False), gives the same NaNs
    np.random.seed(5); torch.manual_seed(5)
    loop = torch.cat([torch.cat([fr.sample_full_swag(Xp) for Xp in torch.chunk(Xb, 10)])[None] for _ in range(3)])
    assert torch.equal(torch.nan_to_num(loop, nan=-1.0), torch.nan_to_num(outs["one", "torch"], nan=-1.0))
    # streamed bands: the simulations (trios) that hold a damaged row get NaN bands, the others the clean batch's
    res = {}
    for name, devs, xx, kw in (("bad1", [0], Xb, {}), ("bad3", [0, 0, 0], Xb, {}), ("clean", [0], X, dict(assume_finite=True))):
        np.random.seed(9)
        res[name] = fr.predictive_bands(xx, samples=32, chunks=10, trios=3, philox_seed=3, samples_per_launch=8, devices=devs, **kw)
    sims_hurt = sorted({b // 3 for b in hurt})
    sims_keep = [s for s in range(30) if s not in sims_hurt]
    p1, p3, pc = (res[k]["percentiles"] for k in ("bad1", "bad3", "clean"))
    assert torch.equal(torch.nan_to_num(p1, nan=-1.0), torch.nan_to_num(p3, nan=-1.0))
    assert torch.isnan(p1[sims_hurt]).all() and torch.equal(p1[sims_keep], pc[sims_keep])
