#!/usr/bin/env python3
"""Golden vectors for the shapes the reference accepts beyond the pretrained ensemble's: the network is built from hparams
(spock_reg_model.py:301-321 mlp(), :346-362: `hidden`, `latent`, depth `in` / `out`, `include_derivatives` doubling the features),
the time pool takes whatever series length it is given (:416-435), and the SWAG rank K comes from swa_params (:700-706).

Every case is the UNMODIFIED reference class (imported through make_golden.import_reference) built from the v50 hparams with the
named keys changed, under its own seed_everything; it gets a synthetic, seeded SWAG state (w_avg = its random init, small positive
variances with one negative element, K deviation columns) -- no pretrained checkpoint has these shapes -- and is run through
forward_swag_fast (:878-908) and forward(noisy_val=False / True) (:486-528) with every random draw taped.  The series-length cases
run the REAL pretrained member v50_0 on x[:, :T].  Build container only.

    python tests/golden/make_golden_arch.py      # writes case_arch_<name>.npz and case_arch_tlen.npz
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import Tape, import_reference, pretrained, save  # noqa: E402

# name -> (hparams overrides, swa_params overrides, systems)
CASES = {
    "h64l16": (dict(hidden=64, latent=16), dict(), 16),                    # parse_swag_args.py:11-12 --hidden 64 --latent 16
    "h20l10": (dict(hidden=20, latent=10), dict(), 16),                    # latent not a multiple of 4
    "h33l7": (dict(hidden=33, latent=7), dict(), 12),                      # neither width a multiple of 4
    "deep22": (dict(**{"in": 2, "out": 2}), dict(), 16),                   # two hidden->hidden layers in each MLP
    "deep30": (dict(hidden=24, latent=12, **{"in": 3, "out": 0}), dict(), 12),   # out = 0: regress_nn is ONE Linear(2L, 2)
    "lin00": (dict(latent=8, **{"in": 0, "out": 0}), dict(), 12),          # in = 0: feature_nn is ONE Linear(41, L)
    "deriv82": (dict(include_derivatives=True), dict(), 8),                # n_features = 82 (:346, :358)
    "k40": (dict(), dict(K=40), 8),                                         # SWAG rank above 32
    "h48megno": (dict(hidden=48, latent=12, fix_megno=True, fix_megno2=False), dict(), 12),
    "h128l32": (dict(hidden=128, latent=32, lower_std=True), dict(K=6), 8),
    "allcols": (dict(hidden=16, latent=4, include_mmr=True, include_nan=True, include_eplusminus=True, fix_megno2=False), dict(K=5), 8),
    "lin0out8": (dict(hidden=24, latent=8, **{"in": 0, "out": 8}), dict(K=8), 8),   # regress_nn alone holds 10 of the 11 Linear modules
}


def typed_hparams(z):
    hp = json.loads(str(z["hparams_json"]))
    for k, v in list(hp.items()):   # the json round trip stringified non-scalars; the reference only needs these keys typed
        if isinstance(v, str) and v in ("True", "False"):
            hp[k] = v == "True"
    return hp


def inputs(B, F, g):
    x = np.load(os.path.join(HERE, "inputs.npz"))["x_slow"][:B].copy()      # [B,100,41], the "slow" distribution
    x[:, :, 7] = (2.0 + 0.5 * torch.randn(B, 100, generator=g) + torch.linspace(0, 1, 100)[None]).numpy()   # a MEGNO column with structure
    if F == 82:   # "derivatives": finite differences along time, scaled to unit order
        dx = np.gradient(x, axis=1) * 10.0
        x = np.concatenate([x, dx.astype(np.float32)], axis=2)
    return torch.tensor(np.ascontiguousarray(x, dtype=np.float32))


def run_case(srm, name, hp_over, swa_over, B, z0):
    hp = typed_hparams(z0)
    hp.update(hp_over)
    hp["seed"] = 7000 + sum(map(ord, name))
    swa = json.loads(str(z0["swa_params_json"]))
    swa.update(swa_over)
    m = srm.SWAGModel(dict(hp)).init_params(dict(swa)).cpu()
    m.eval()
    d = m.flatten().numel()
    g = torch.Generator().manual_seed(hp["seed"] + 1)
    w0 = m.flatten().detach().clone()
    m.w_avg = w0.clone()
    m.w2_avg = w0 ** 2 + (0.02 * torch.rand(d, generator=g)) ** 2
    m.w2_avg[d // 3] = w0[d // 3] ** 2 - 1e-6                      # one negative variance element, like 5 of the 30 pretrained members
    m.pre_D = (w0[:, None] + 0.05 * torch.randn(d, m.K, generator=g)).contiguous()
    x = inputs(B, m.n_features, g)
    out = {}
    torch.manual_seed(hp["seed"] + 2)
    with Tape() as tape:
        o = m.forward_swag_fast(x, scale=0.5).detach()
    out.update(swagfast_out=o.numpy(), swagfast_w=m.flatten().detach().numpy().copy(), **tape.as_dict("swagfast_tape"))
    w_loaded = m.flatten().detach().clone()
    for noisy in (False, True):
        torch.manual_seed(hp["seed"] + 3 + int(noisy))
        with Tape() as tape:
            o = m(x, noisy_val=noisy).detach()
        out[f"forward_noisy{int(noisy)}_out"] = o.numpy()
        out[f"forward_noisy{int(noisy)}_summary"] = m._cur_summary.detach().numpy()   # before summary noise
        out.update(tape.as_dict(f"forward_noisy{int(noisy)}_tape"))
    assert torch.equal(m.flatten().detach(), w_loaded)
    hp_out = {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in dict(m.hparams).items()}
    sd = m.state_dict()
    save(f"case_arch_{name}.npz", x=x.numpy(), w_avg=m.w_avg.numpy(), w2_avg=m.w2_avg.numpy(), pre_D=m.pre_D.numpy(),
         hparams_json=np.array(json.dumps(hp_out)), swa_params_json=np.array(json.dumps(dict(m.swa_params))),
         state_keys=np.array(list(sd.keys())), state_sizes=np.array([v.numel() for v in sd.values()]),
         state_shapes=np.array(json.dumps([list(v.shape) for v in sd.values()])), n_features=np.array(m.n_features), **out)


def run_tlen(srm):
    """Series lengths other than 100 through the real pretrained member v50_0 (the pool takes any T, :416-435; torch.std needs T >= 2)."""
    m = srm.load_swag(pretrained(0)).cpu()
    m.eval()
    x = torch.tensor(np.load(os.path.join(HERE, "inputs.npz"))["x_slow"][:16].copy())
    out = {}
    for T in (2, 3, 5, 6, 7, 99):
        xt = x[:, :T].contiguous()
        torch.manual_seed(8800 + T)
        with Tape() as tape:
            o = m.forward_swag_fast(xt, scale=0.5).detach()
        out[f"T{T}_out"] = o.numpy()
        out[f"T{T}_w"] = m.flatten().detach().numpy().copy()
        out.update(tape.as_dict(f"T{T}_tape"))
        w = m.flatten().detach().clone()
        torch.manual_seed(8900 + T)
        with Tape() as tape:
            o = m(xt, noisy_val=True).detach()
        out[f"T{T}_noisy_out"] = o.numpy()
        out.update(tape.as_dict(f"T{T}_noisy_tape"))
        assert torch.equal(m.flatten().detach(), w)
    save("case_arch_tlen.npz", x=x.numpy(), lengths=np.array([2, 3, 5, 6, 7, 99]), **out)


def main():
    srm = import_reference()
    torch.set_num_threads(1)
    z0 = np.load(os.path.join(HERE, "swag_v50_0.npz"))
    only = sys.argv[1:]
    for name, (hp_over, swa_over, B) in CASES.items():
        if not only or name in only:
            run_case(srm, name, hp_over, swa_over, B, z0)
    if not only or "tlen" in only:
        run_tlen(srm)


if __name__ == "__main__":
    main()
