#!/usr/bin/env python3
"""Golden vectors for the fix_megno=True branch of the reference (spock_reg_model.py:360-362, 480-484, 488-491, 509-510): the
summary gains [mean_t, std_t] of the RAW MEGNO column, regress_nn.0 and summary_noise_logvar are two wider (d = 7665).

No pretrained checkpoint has fix_megno=True, so the model is built by the UNMODIFIED reference class from the v50 hparams with
that one flag flipped (random init under its own seed_everything), given a synthetic SWAG state (w_avg = its init, small positive
variance, K = 30 deviation columns, all seeded), and run through forward_swag_fast and forward(noisy_val=True / False) with every
random draw taped.  Build container only (imports /root/reference through make_golden.import_reference).

    python tests/golden/make_golden_megno.py      # writes case_megno.npz
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import Tape, import_reference, save  # noqa: E402


def main():
    srm = import_reference()
    torch.set_num_threads(1)
    z0 = np.load(os.path.join(HERE, "swag_v50_0.npz"))
    hp = json.loads(str(z0["hparams_json"]))
    for k, v in list(hp.items()):       # the json round trip stringified non-scalars; the reference only needs these keys typed
        if isinstance(v, str) and v in ("True", "False"):
            hp[k] = v == "True"
    hp["fix_megno"] = True
    hp["fix_megno2"] = False
    hp["seed"] = 4242
    swa = json.loads(str(z0["swa_params_json"]))
    m = srm.SWAGModel(dict(hp)).init_params(dict(swa)).cpu()
    m.eval()
    assert m.fix_megno and m.regress_nn[0].weight.shape == (40, 42) and m.summary_noise_logvar.numel() == 42
    d = m.flatten().numel()
    assert d == 7665, d
    g = torch.Generator().manual_seed(99)
    w0 = m.flatten().detach().clone()
    m.w_avg = w0.clone()
    m.w2_avg = w0 ** 2 + (0.02 * torch.rand(d, generator=g)) ** 2
    m.w2_avg[17] = w0[17] ** 2 - 1e-6                       # one negative variance element, like 5 of the 30 pretrained members
    m.pre_D = (w0[:, None] + 0.05 * torch.randn(d, m.K, generator=g)).contiguous()
    x = np.load(os.path.join(HERE, "inputs.npz"))["x_slow"][:16].copy()
    x[:, :, 7] = (2.0 + 0.5 * torch.randn(16, 100, generator=g) + torch.linspace(0, 1, 100)[None]).numpy()  # a MEGNO column with structure
    x = torch.tensor(x)
    out = {}
    # forward_swag_fast (:878-908)
    torch.manual_seed(5150)
    with Tape() as tape:
        o = m.forward_swag_fast(x, scale=0.5).detach()
    out.update(swagfast_out=o.numpy(), swagfast_w=m.flatten().detach().numpy().copy(), **tape.as_dict("swagfast_tape"))
    # forward with the sampled weights loaded: quiet and noisy (:486-528)
    w_loaded = m.flatten().detach().clone()
    for noisy in (False, True):
        torch.manual_seed(5151 + int(noisy))
        with Tape() as tape:
            o = m(x, noisy_val=noisy).detach()
        out[f"forward_noisy{int(noisy)}_out"] = o.numpy()
        out[f"forward_noisy{int(noisy)}_summary"] = m._cur_summary.detach().numpy()   # before summary noise, 42 wide
        out.update(tape.as_dict(f"forward_noisy{int(noisy)}_tape"))
    assert torch.equal(m.flatten().detach(), w_loaded)
    hp_out = {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in dict(m.hparams).items()}
    save("case_megno.npz", x=x.numpy(), w_avg=m.w_avg.numpy(), w2_avg=m.w2_avg.numpy(), pre_D=m.pre_D.numpy(),
         hparams_json=np.array(json.dumps(hp_out)), swa_params_json=np.array(json.dumps(dict(m.swa_params))),
         state_keys=np.array(list(m.state_dict().keys())), state_sizes=np.array([v.numel() for v in m.state_dict().values()]), **out)


if __name__ == "__main__":
    main()
