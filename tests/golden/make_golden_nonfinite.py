#!/usr/bin/env python3
"""Golden vectors for NON-FINITE inputs: what the unmodified reference returns when x holds NaN / +inf / -inf.

The reference masks by subtraction -- `x = x - mask` (spock_reg_model.py:452-478) -- so a non-finite value in a MASKED column becomes
NaN (NaN - NaN, inf - inf) instead of 0; nn.Linear multiplies it into every neuron (NaN x 0 = NaN too) and nn.ReLU (:301-321)
propagates NaN, so the whole system's (mu, std) is NaN.  In a LIVE column NaN does the same; +-inf becomes +-inf x weight, ReLU keeps
+inf and turns -inf into 0, and the next Linear mixes +inf and -inf into NaN -- unless every weight the infinity meets has one sign.
The `dead` case below builds such a network with the reference's own class (feature_nn.0's weights on one live column all negative and
exactly known: zero SWAG variance on them), so that +inf on that column is killed by the ReLU and the reference's outputs stay FINITE,
while -inf on it gives NaN: the one input class where "non-finite in, NaN out" is not the whole story.

Cases: the real pretrained members v50_0 and v50_12 through forward_swag_fast (:878-908) and forward(noisy_val=False / True)
(:486-528), plus the `dead` network (hidden 20, latent 10) through the same three calls; every random draw taped.  Build container only.

    python tests/golden/make_golden_nonfinite.py      # writes case_nonfinite.npz
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import Tape, import_reference, pretrained, save  # noqa: E402
from make_golden_arch import typed_hparams  # noqa: E402

NAN, PINF, NINF = float("nan"), float("inf"), float("-inf")
NEG_NAN = np.frombuffer(np.uint32(0xFFC00000).tobytes(), np.float32)[0]   # what x86 produces for 0/0: the sign bit set


def poison(x):
    """x [>=20,100,41] -> a copy with one kind of damage per system (system 0 stays clean) and the list of what was done."""
    x = x.copy()
    what = ["clean"]
    def put(b, t, c, v, label):
        x[b, t, c] = v
        what.append(label)
    put(1, 17, 9, NAN, "NaN, live column 9, t=17")
    put(2, 3, 12, PINF, "+inf, live column 12, t=3")
    put(3, 50, 20, NINF, "-inf, live column 20, t=50")
    put(4, 0, 3, NAN, "NaN, masked column 3 (mmr), t=0")
    put(5, 99, 7, PINF, "+inf, masked column 7 (megno), t=99")
    put(6, 10, 38, NINF, "-inf, masked column 38 (nan flags), t=10")
    put(7, 5, 1, NAN, "NaN, masked column 1 (eplusminus), t=5")
    put(8, slice(None), 6, PINF, "+inf, masked column 6, every timestep")
    put(9, 42, slice(None), NAN, "NaN, the whole row t=42")
    put(10, 0, slice(None), PINF, "+inf, the whole row of the first timestep")
    put(11, 99, slice(None), NINF, "-inf, the whole row of the last timestep")
    put(12, 0, 0, NAN, "NaN, the time column, first timestep")
    put(13, 99, 37, PINF, "+inf, live column 37, last timestep")
    put(14, 33, 15, NEG_NAN, "NaN with the sign bit set, live column 15")
    put(15, 34, 40, NEG_NAN, "NaN with the sign bit set, masked column 40")
    put(16, slice(None), 25, NINF, "-inf, live column 25, every timestep")
    put(17, 60, 2, NINF, "-inf, masked column 2, t=60")
    x[18, 7, 5] = PINF; x[18, 8, 10] = NAN
    what.append("+inf masked column 5 and NaN live column 10")
    x[19, 11, 8] = 1e30
    what.append("finite 1e30, live column 8 (stays finite: saturated)")
    return x, what


def three_calls(m, x, seed, prefix, out):
    torch.manual_seed(seed)
    with Tape() as tape:
        o = m.forward_swag_fast(x, scale=0.5).detach()
    out[f"{prefix}_swagfast_out"] = o.numpy()
    out[f"{prefix}_swagfast_w"] = m.flatten().detach().numpy().copy()
    out.update(tape.as_dict(f"{prefix}_swagfast_tape"))
    w = m.flatten().detach().clone()
    for noisy in (False, True):
        torch.manual_seed(seed + 1 + int(noisy))
        with Tape() as tape:
            o = m(x, noisy_val=noisy).detach()
        out[f"{prefix}_forward_noisy{int(noisy)}_out"] = o.numpy()
        out[f"{prefix}_forward_noisy{int(noisy)}_summary"] = m._cur_summary.detach().numpy()
        out[f"{prefix}_forward_noisy{int(noisy)}_latents"] = m.latents.detach().numpy()[:4]     # [4,T,L] self.latents (:433)
        out.update(tape.as_dict(f"{prefix}_forward_noisy{int(noisy)}_tape"))
    assert torch.equal(m.flatten().detach(), w)


def main():
    srm = import_reference()
    torch.set_num_threads(1)
    base = np.load(os.path.join(HERE, "inputs.npz"))["x_slow"][:20]
    xp, what = poison(base)
    x = torch.tensor(xp)
    out = {}
    for si in (0, 12):
        m = srm.load_swag(pretrained(si)).cpu()
        m.eval()
        three_calls(m, x, 9100 + 10 * si, f"v50_{si}", out)
    # VarModel.sample (:530-545) on the poisoned batch: np.average over mu + randn * std
    m = srm.load_swag(pretrained(0)).cpu()
    m.eval()
    m.load(m.w_avg)
    torch.manual_seed(9300)
    np.random.seed(9300)
    with Tape() as tape:
        s = m.sample(x, samples=2)
    out.update(sample_out=np.asarray(s), sample_w=m.flatten().detach().numpy(), **tape.as_dict("sample_tape"))

    # ---- the `dead` network: +inf on live column DEADCOL meets only negative weights in feature_nn.0 and dies in the ReLU
    DEADCOL = 9
    z0 = np.load(os.path.join(HERE, "swag_v50_0.npz"))
    hp = typed_hparams(z0)
    hp.update(hidden=20, latent=10, seed=7717)
    swa = json.loads(str(z0["swa_params_json"]))
    swa.update(K=6)
    md = srm.SWAGModel(dict(hp)).init_params(dict(swa)).cpu()
    md.eval()
    sd = md.state_dict()
    keys = list(sd.keys())
    off = {}
    o = 0
    for k in keys:
        off[k] = o
        o += sd[k].numel()
    d = o
    g = torch.Generator().manual_seed(7718)
    w0 = md.flatten().detach().clone()
    W1 = w0[off["feature_nn.0.weight"]:off["feature_nn.0.weight"] + 20 * 41].view(20, 41)
    W1[:, DEADCOL] = -(0.05 + 0.2 * torch.rand(20, generator=g))          # every weight on the column negative
    md.w_avg = w0.clone()
    md.w2_avg = w0 ** 2 + (0.02 * torch.rand(d, generator=g)) ** 2
    md.pre_D = (w0[:, None] + 0.05 * torch.randn(d, md.K, generator=g)).contiguous()
    dead_idx = off["feature_nn.0.weight"] + torch.arange(20) * 41 + DEADCOL
    md.w2_avg[dead_idx] = w0[dead_idx] ** 2                                # zero variance and zero deviation there: every draw keeps them
    md.pre_D[dead_idx] = w0[dead_idx, None]
    xd = base[:8].copy()
    xd[1, 20, DEADCOL] = PINF          # dies in the ReLU: finite outputs
    xd[2, 21, DEADCOL] = NINF          # +inf after feature_nn.0: NaN
    xd[3, :, DEADCOL] = PINF           # every timestep: still finite
    xd[4, 5, DEADCOL] = NAN
    xd[5, 6, DEADCOL] = PINF; xd[5, 6, 3] = 2.5     # a finite value in a masked column next to it: no effect
    xd[6, 7, DEADCOL] = PINF; xd[6, 9, 3] = PINF    # ... a non-finite one: NaN
    xd = torch.tensor(xd)
    three_calls(md, xd, 9400, "dead", out)
    assert np.isfinite(out["dead_swagfast_out"][[0, 1, 3, 5, 7]]).all() and np.isnan(out["dead_swagfast_out"][[2, 4, 6]]).all()
    hp_out = {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in dict(md.hparams).items()}
    save("case_nonfinite.npz", x=xp, what=np.array(what), dead_x=xd.numpy(), dead_col=np.array(DEADCOL),
         dead_w_avg=md.w_avg.numpy(), dead_w2_avg=md.w2_avg.numpy(), dead_pre_D=md.pre_D.numpy(),
         dead_hparams_json=np.array(json.dumps(hp_out)), dead_swa_params_json=np.array(json.dumps(dict(md.swa_params))), **out)
    for k in ("v50_0_swagfast_out", "v50_0_forward_noisy0_out", "v50_0_forward_noisy1_out", "v50_12_swagfast_out", "dead_swagfast_out",
              "dead_forward_noisy1_out", "sample_out"):
        v = out[k]
        print(k, "NaN rows:", np.where(np.isnan(v).reshape(len(v), -1).any(1))[0].tolist(),
              "inf rows:", np.where(np.isinf(v).reshape(len(v), -1).any(1))[0].tolist())


if __name__ == "__main__":
    main()
