#!/usr/bin/env python3
"""Golden-vector generator for the MultiSWAG inference hot path.

Runs ONLY in the build container: it imports the UNMODIFIED reference file
/root/reference/spock_reg_model.py (through a stub for the two modules this image
lacks, pytorch_lightning and torch._six -- SURVEY.md Appendix A), replays the
reference's own code on seeded inputs, and writes small .npz fixtures next to this
script.  The reference never travels to the GPU box; only these fixtures do.

Every random number the reference consumes is captured on a "tape" by wrapping
torch.randn / torch.randn_like / np.random.randint / np.random.randn, so the
fixtures hold the exact noise in the reference's own consumption order
 without this script restating that order.

    python tests/golden/make_golden.py            # regenerates every fixture

Fixtures (all float32 unless noted):
  swag_v50_0.npz, swag_v50_12.npz   converted SWAG state (w_avg, w2_avg, pre_D), hparams json,
                                    ssX mean/scale (float64)
  inputs.npz                        x_slow[32,100,41], x_iid[8,100,41], x_const4[4,100,41]
  case_*.npz                        one reference call each: inputs named, tape, expected outputs
"""
import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# --------------------------------------------------------------------------- shim
def import_reference():
    six = types.ModuleType("torch._six")
    six.inf = float("inf")
    sys.modules["torch._six"] = six
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    def seed_everything(s):
        import random
        random.seed(s)
        np.random.seed(s)
        torch.manual_seed(s)
        return s

    class Trainer:
        pass

    pl.LightningModule, pl.seed_everything, pl.Trainer = LightningModule, seed_everything, Trainer
    util = types.ModuleType("pytorch_lightning.utilities")
    parsing = types.ModuleType("pytorch_lightning.utilities.parsing")

    class AttributeDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    AttributeDict.__module__ = "pytorch_lightning.utilities.parsing"
    parsing.AttributeDict = AttributeDict
    util.parsing = parsing
    pl.utilities = util
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.utilities": util,
                        "pytorch_lightning.utilities.parsing": parsing})
    sys.path.insert(0, REF)
    _tl = torch.load
    torch.load = lambda p, *a, **k: _tl(p, *a, **{**k, "weights_only": False})
    import spock_reg_model  # the unmodified reference file
    return spock_reg_model


# --------------------------------------------------------------------------- RNG tape
class Tape:
    """Records every normal / integer draw the reference makes, in order."""

    def __init__(self):
        self.items = []
        self._orig = {}

    def __enter__(self):
        self.items = []
        o = self._orig
        o["randn"], o["randn_like"] = torch.randn, torch.randn_like
        o["np_randint"], o["np_randn"] = np.random.randint, np.random.randn

        def randn(*a, **k):
            r = o["randn"](*a, **k)
            self.items.append(("torch.randn", r.detach().cpu().numpy().copy()))
            return r

        def randn_like(*a, **k):
            r = o["randn_like"](*a, **k)
            self.items.append(("torch.randn_like", r.detach().cpu().numpy().copy()))
            return r

        def np_randint(*a, **k):
            r = o["np_randint"](*a, **k)
            self.items.append(("np.randint", np.asarray(r).copy()))
            return r

        def np_randn(*a, **k):
            r = o["np_randn"](*a, **k)
            self.items.append(("np.randn", np.asarray(r).copy()))
            return r

        torch.randn, torch.randn_like = randn, randn_like
        np.random.randint, np.random.randn = np_randint, np_randn
        return self

    def __exit__(self, *exc):
        o = self._orig
        torch.randn, torch.randn_like = o["randn"], o["randn_like"]
        np.random.randint, np.random.randn = o["np_randint"], o["np_randn"]

    def as_dict(self, prefix="tape"):
        d = {f"{prefix}_n": np.array(len(self.items))}
        kinds = []
        for i, (kind, arr) in enumerate(self.items):
            d[f"{prefix}_{i:03d}"] = arr
            kinds.append(kind)
        d[f"{prefix}_kinds"] = np.array(kinds)
        return d


# --------------------------------------------------------------------------- inputs
def make_inputs():
    g = torch.Generator().manual_seed(123)
    B = 32
    base = torch.randn(B, 1, 41, generator=g)
    n = torch.randn(B, 100, 41, generator=g)
    x_slow = base + 0.1 * n
    x_slow[:, :, 0] = torch.linspace(-1.71, 1.74, 100)[None]
    x_iid = torch.randn(8, 100, 41, generator=g)
    return x_slow.float().contiguous(), x_iid.float().contiguous()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


def pretrained(i):
    import glob
    fs = glob.glob(f"{REF}/pretrained/*_v50_{i}_output.pkl")
    assert len(fs) == 1, fs
    return fs[0]


def masked(m, x):
    """The reference's masking block (spock_reg_model.py:884-897), by calling its own methods."""
    if m.fix_megno or m.fix_megno2:
        x = m.zero_megno(x)
    if not m.include_mmr:
        x = m.zero_mmr(x)
    if not m.include_nan:
        x = m.zero_nan(x)
    if not m.include_eplusminus:
        x = m.zero_eplusminus(x)
    return x


def main():
    srm = import_reference()
    torch.set_num_threads(1)  # fixed summation order inside MKL for reproducible fixtures
    x_slow, x_iid = make_inputs()

    models = {}
    for i in (0, 12):
        m = srm.load_swag(pretrained(i)).cpu()
        m.eval()
        models[i] = m
        hp = {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in dict(m.hparams).items()}
        save(f"swag_v50_{i}.npz",
             w_avg=m.w_avg.numpy(), w2_avg=m.w2_avg.numpy(), pre_D=m.pre_D.numpy(),
             hparams_json=np.array(json.dumps(hp)), swa_params_json=np.array(json.dumps(dict(m.swa_params))),
             ssX_mean=m.ssX.mean_, ssX_scale=m.ssX.scale_,
             state_keys=np.array(list(m.state_dict().keys())),
             state_sizes=np.array([v.numel() for v in m.state_dict().values()]))

    # ---- case I: constructor side effects (spock_reg_model.py:343-345, 359-362): seed_everything(seed) + module init.
    # After load_swag the module holds its RANDOM init and the global generators sit in a seed-determined state.
    mi = srm.load_swag(pretrained(3)).cpu()
    save("case_init_v50_3.npz", seed=np.array(mi.hparams["seed"]), init_flat=mi.flatten().detach().numpy(),
         next_torch=torch.randn(4).numpy(), next_numpy=np.random.rand(4), w_avg_head=mi.w_avg[:8].numpy(),
         hparams_json=np.array(json.dumps({k: (v if isinstance(v, (int, float, str, bool)) else str(v))
                                           for k, v in dict(mi.hparams).items()})),
         swa_params_json=np.array(json.dumps(dict(mi.swa_params))))

    # ---- case F: feature packing.  figures/spock/regression.py is not importable here (rebound, numba, ...), but
    # data_setup_kernel (:183-213) is pure numpy: its source is cut out of the file with ast and executed as is
    # (the @jit decorator and numpy<1.24's np.float alias are supplied by the namespace).
    import ast
    src = open(f"{REF}/figures/spock/regression.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "data_setup_kernel"][0]
    npx = types.SimpleNamespace(**{k: getattr(np, k) for k in dir(np) if not k.startswith("__")})
    npx.float = float
    ns = {"np": npx, "jit": lambda f: f}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "regression.py:data_setup_kernel", "exec"), ns)
    rng = np.random.default_rng(77)
    ts = rng.standard_normal((6, 1, 100, 26)) * 2.0
    ts[0, 0, 5, 3] = np.nan; ts[1, 0, 7, 6] = np.inf; ts[2, 0, 9, 7] = -np.inf; ts[3, 0, 11, 12] = np.nan; ts[4, 0, 0, 0] = np.inf
    masses = np.abs(rng.standard_normal((6, 3))) * 1e-5
    Xs = np.stack([ns["data_setup_kernel"](masses[i], ts[i])[0] for i in range(6)])           # [6,100,41]
    ss = models[0].ssX
    Xp = ss.transform(Xs.reshape(-1, 41)).reshape(Xs.shape)
    save("case_features.npz", tseries=ts[:, 0], mass=masses, X64=Xs, x32=torch.tensor(Xp).float().numpy(),
         mean=ss.mean_, scale=ss.scale_)

    # ---- case S: the post-sampling statistics of figures/multiswag_5_planet.py.  The script cannot
    # run here (rebound, xgboost, ...), so the function fast_truncnorm (:306-370) and the top-level statements of the
    # prior-resampling block (:396-422) and of the min over trios (:428) are cut out with ast and executed as they are.
    src5 = open(f"{REF}/figures/multiswag_5_planet.py").read()
    mod5 = ast.parse(src5)
    ftn = [n for n in mod5.body if isinstance(n, ast.FunctionDef) and n.name == "fast_truncnorm"][0]
    ns5 = {"np": np, "jnp": np}
    exec(compile(ast.Module(body=[ftn], type_ignores=[]), "multiswag_5_planet.py:fast_truncnorm", "exec"), ns5)
    rng = np.random.default_rng(99)
    samples_, sims_ = 7, 40
    loc = rng.uniform(3.0, 12.5, size=(samples_, sims_, 3)).astype(np.float32)   # "time[..., 0]"
    loc[0, :5] = 3.2                                                               # hard cases: most candidates below 4
    scale = rng.uniform(0.5, 6.0, size=(samples_, sims_, 3)).astype(np.float32)
    scale[0, :5] = 0.05                                                            # ... and no candidate passes
    np.random.seed(6000)
    with Tape() as tape_tn:
        o_normal = np.random.normal
        def rec_normal(*a, **k):
            r_ = o_normal(*a, **k)
            tape_tn.items.append(("np.normal", np.asarray(r_).copy()))
            return r_
        np.random.normal = rec_normal
        try:
            samps_time = np.array(ns5["fast_truncnorm"](loc, scale, left=4, d=300, nsamp=40, seed=0))
        finally:
            np.random.normal = o_normal
    assert samps_time.dtype == np.float32
    trunc = samps_time.copy()
    blk = [n for n in mod5.body if 396 <= n.lineno <= 422]
    ns6 = {"np": np, "samps_time": samps_time}
    np.random.seed(6001)
    o_rand = np.random.rand
    rec = []
    np.random.rand = lambda *a: (rec.append(o_rand(*a)) or rec[-1])
    try:
        exec(compile(ast.Module(body=blk, type_ignores=[]), "multiswag_5_planet.py:396-422", "exec"), ns6)
    finally:
        np.random.rand = o_rand
    outs_ = np.min(ns6["samps_time"], 2).T                                         # :428
    save("case_stats.npz", loc=loc, scale=scale, left=np.array(4.0), nsamp=np.array(40), d=np.array(300),
         truncnorm=trunc, resampled=ns6["samps_time"], u=rec[0], n_samples=np.array(int(ns6["n_samples"])),
         normalization=np.array(ns6["normalization"]), cum_values=np.array(ns6["cum_values"], dtype=np.float64),
         bin_edges=np.array(ns6["bin_edges"], dtype=np.float64), outs=outs_, **tape_tn.as_dict("normals"))

    m0 = models[0]
    # constant-4 "unstable" fill (figures/multiswag_5_planet.py:214-215) after ssX, float64 transform then .float()
    raw4 = np.ones((4, 100, 41)) * 4
    x_const4 = torch.tensor(m0.ssX.transform(raw4.reshape(-1, 41)).reshape(raw4.shape)).float()
    save("inputs.npz", x_slow=x_slow.numpy(), x_iid=x_iid.numpy(), x_const4=x_const4.numpy())

    # ---- case A: forward_swag_fast(x, 0.5), with intermediates, for both seeds and three inputs
    for si, m in models.items():
        for xname, x in (("slow", x_slow), ("iid", x_iid), ("const4", x_const4)):
            torch.manual_seed(1000 + si)
            with Tape() as tape:
                out = m.forward_swag_fast(x, scale=0.5).detach()
            w = m.flatten().detach().clone()
            # intermediates: replay by hand with the SAME weights and the taped pooling noise
            xm = masked(m, x)
            with torch.no_grad():
                lat = m.feature_nn(xm)
            # second run of the whole call to confirm replay determinism
            torch.manual_seed(1000 + si)
            out2 = m.forward_swag_fast(x, scale=0.5).detach()
            assert torch.equal(out, out2)
            # summary + pre-clamp by replaying compute_summary_stats with the taped noise
            torch.manual_seed(1000 + si)
            m.sample_weights(scale=0.5)
            with torch.no_grad():
                summ = m.compute_summary_stats(xm)
                pre = m.regress_nn(summ)
                mu, std = m.predict_instability(summ)
            assert torch.equal(torch.cat((mu, std), 1), out)
            # fp64 truth of the same math and the same noise
            md = srm.SWAGModel(dict(m.hparams)).init_params(dict(m.swa_params)).double()
            md.w_avg, md.w2_avg, md.pre_D = m.w_avg.double(), m.w2_avg.double(), m.pre_D.double()
            z1 = torch.tensor(tape.items[0][1]).double()
            z2 = torch.tensor(tape.items[1][1]).double()
            e1 = torch.tensor(tape.items[2][1]).double()
            e2 = torch.tensor(tape.items[3][1]).double()
            D = md.pre_D - md.w_avg[:, None]
            wd = md.w_avg + 0.5 / np.sqrt(2.0) * z1[0] * torch.sqrt(torch.abs(md.w2_avg - md.w_avg ** 2))
            wd = wd + 0.5 * (D @ z2)[:, 0] / np.sqrt(2 * (md.K - 1))
            md.load(wd)
            with torch.no_grad():
                latd = md.feature_nn(masked(md, x.double()))
                smu = latd.mean(1)
                svar = latd.std(1) ** 2
                n = latd.shape[1]
                mus = e1 * torch.sqrt(svar / n) + smu
                vas = e2 * torch.sqrt(2 * svar ** 2 / (n - 1)) + svar
                summd = torch.cat((mus, torch.sqrt(torch.abs(vas) + srm.EPSILON)), 1)
                pred = md.regress_nn(summd)
                mud, stdd = md.predict_instability(summd)
            save(f"case_swagfast_v50_{si}_{xname}.npz",
                 scale=np.array(0.5), torch_seed=np.array(1000 + si), out=out.numpy(), w=w.numpy(),
                 latents=lat[:2].numpy(), summary=summ.numpy(), pre_clamp=pre.numpy(),
                 out_f64=torch.cat((mud, stdd), 1).numpy(), pre_clamp_f64=pred.numpy(), w_f64=wd.numpy(),
                 **tape.as_dict())

    # ---- case B: VarModel.forward with loaded weights w_avg (noisy_val False / True)
    for si, m in models.items():
        m.load(m.w_avg)
        for noisy in (False, True):
            torch.manual_seed(2000 + si)
            with Tape() as tape:
                out = m.forward(x_slow, noisy_val=noisy).detach()
            save(f"case_forward_v50_{si}_noisy{int(noisy)}.npz", torch_seed=np.array(2000 + si),
                 out=out.numpy(), w=m.flatten().detach().numpy(), **tape.as_dict())

    # ---- case C: VarModel.sample(x, samples=3) (spock_reg_model.py:530-545)
    m0.load(m0.w_avg)
    torch.manual_seed(3000)
    np.random.seed(3000)
    with Tape() as tape:
        s = m0.sample(x_slow, samples=3)
    save("case_sample_v50_0.npz", samples=np.array(3), out=np.asarray(s), w=m0.flatten().detach().numpy(),
         **tape.as_dict())

    # ---- case D: FeatureRegressor.sample_full_swag semantics (figures/spock/regression.py:74-92 is not
    # importable here: needs rebound/numba/...). Its 19 lines are: seed pick with np.random.randint(0, S),
    # .eval(), forward_swag_fast(X, scale=0.5).  Run that around the imported SWAGModel objects.
    ensemble = [models[0], models[12]]

    def sample_full_swag(X):
        swag_i = np.random.randint(0, len(ensemble))
        mm = ensemble[swag_i]
        mm.eval()
        return mm.forward_swag_fast(X, scale=0.5)

    np.random.seed(4000)
    torch.manual_seed(4000)
    with Tape() as tape:
        outs = torch.cat([sample_full_swag(x_slow)[None].detach() for _ in range(3)])
    save("case_multiswag_grid.npz", out=outs.numpy(), ensemble=np.array([0, 12]), **tape.as_dict())

    # ---- case E: the 5-planet MC driver loop (figures/multiswag_5_planet.py:295-298): samples x chunks
    Xflat = x_slow[:30]
    samples = 2
    np.random.seed(5000)
    torch.manual_seed(5000)
    with Tape() as tape:
        time = torch.cat([
            torch.cat([sample_full_swag(Xpart).detach().cpu() for Xpart in torch.chunk(Xflat, chunks=10)])[None]
            for _ in range(samples)], dim=0)
    save("case_chunk_loop.npz", out=time.numpy(), ensemble=np.array([0, 12]), chunks=np.array(10),
         samples=np.array(samples), nrows=np.array(30), **tape.as_dict())


if __name__ == "__main__":
    main()
