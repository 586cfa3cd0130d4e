#!/usr/bin/env python3
"""Parity at scale for the networks the reference builds from OTHER hparams (spock_reg_model.py:301-321, 343-397): case_scale_arch.npz.

For every architecture fixture case_arch_<name>.npz with 41 features (make_golden_arch.py: the unmodified reference class built with other
widths / depths / masks / K / fix_megno on a seeded synthetic SWAG state) the SAME reference class is rebuilt from the fixture's hparams
and state and evaluates 2 048 systems through forward_swag_fast(x, 0.5) (:878-908) and 512 through forward(noisy_val=True) (:486-528) at
w_avg -- once in float32 and once with the same code in float64 (the truth).  Inputs and normals come from scale_recipe.py (nothing of
them is stored), handed to the reference by Player in its own consumption order.  Outputs only: ~0.4 MB.  Build container only.

    python tests/golden/make_golden_scale_arch.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import scale_recipe as R  # noqa: E402
from make_golden import import_reference  # noqa: E402
from make_golden_arch import typed_hparams  # noqa: E402
import json  # noqa: E402

NAMES = ("h64l16", "h20l10", "h33l7", "deep22", "deep30", "lin00", "k40", "h48megno", "allcols", "lin0out8")   # (h128l32: its float64 draw needs 21 GB per temporary)
SYSTEMS, NOISY = 2048, 512
BLOCK0 = 40       # recipe "member" index of the first architecture's block of systems (0..29 are the pretrained members')


def build(srm, z, double):
    m = srm.SWAGModel(typed_hparams(z)).init_params(json.loads(str(z["swa_params_json"]))).cpu()
    m.eval()
    cast = (lambda a: torch.tensor(a).double()) if double else torch.tensor
    if double:
        m = m.double()
    m.w_avg, m.w2_avg, m.pre_D = cast(z["w_avg"]), cast(z["w2_avg"]), cast(z["pre_D"])
    return m


def main():
    srm = import_reference()
    torch.set_num_threads(1)
    out = {"names": np.array(NAMES), "systems": np.array(SYSTEMS), "noisy_systems": np.array(NOISY), "block0": np.array(BLOCK0)}
    t0 = time.time()
    for k, name in enumerate(NAMES):
        z = np.load(os.path.join(HERE, f"case_arch_{name}.npz"))
        assert int(z["n_features"]) == 41
        blk = BLOCK0 + k
        x = torch.tensor(R.x_block(blk, 0, SYSTEMS))
        res = {}
        for double in (False, True):
            m = build(srm, z, double)
            d, K, L = m.flatten().numel(), int(m.K), int(m.hparams["latent"])
            SM = 2 * L + (2 if m.fix_megno else 0)
            dt, tdt = (np.float64, torch.float64) if double else (np.float32, torch.float32)
            xx = x.double() if double else x
            with R.Player(R.draw_noise(blk, 0, SYSTEMS, dtype=dt, d=d, k=K, latent=L), tdt), torch.no_grad():
                res[("fast", double)] = m.forward_swag_fast(xx, scale=0.5).numpy()
            m.load(m.w_avg)
            with R.Player(R.noisy_noise(blk, NOISY, dtype=dt, latent=L, summary=SM), tdt), torch.no_grad():
                res[("noisy", double)] = m.forward(xx[:NOISY], noisy_val=True).numpy()
        for leg in ("fast", "noisy"):
            o32, o64 = res[(leg, False)], res[(leg, True)]
            out[f"{name}_{leg}32"] = o32
            out[f"{name}_{leg}_truth_delta"] = (o64 - o32.astype(np.float64)).astype(np.float32)
        rel = np.abs(res[("fast", False)] - res[("fast", True)]) / np.abs(res[("fast", True)])
        print(f"{name:9s} d={d:6d} K={K:2d} latent={L:2d}: reference fp32 vs its float64: max rel {rel.max():.2e}, beyond 1e-5: {(rel > 1e-5).sum()}"
              f"  [{time.time() - t0:.0f} s]", flush=True)
    path = os.path.join(HERE, "case_scale_arch.npz")
    np.savez(path, **out)
    print(f"wrote case_scale_arch.npz: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
