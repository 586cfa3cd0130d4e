#!/usr/bin/env python3
"""Parity AT SCALE against the reference itself: case_scale.npz (build container only).

The UNMODIFIED /root/reference/spock_reg_model.py (through make_golden.import_reference's two-module stub) evaluates

  * all 30 pretrained members x 2 weight draws x the member's own block of 4 096 systems through
    SWAGModel.forward_swag_fast(x, scale=0.5) (:878-908)                       -> 245 760 evaluations, 491 520 outputs;
  * all 30 members at w_avg through VarModel.forward(x, noisy_val=True) (:486-528) on the first 512 systems of the
    block                                                                       ->  15 360 evaluations,  30 720 outputs;

once in float32 (what the scripts run: figures/multiswag_5_planet.py:287) and once with the SAME code in float64 (model.double(),
state and inputs cast up, the same normals) = the truth the two fp32 implementations are measured against.

Nothing of the inputs is stored.  x and every normal come from tests/golden/scale_recipe.py (integer hash + one IEEE rounding:
numpy reproduces them bit for bit anywhere); the reference receives its normals through Player, the playing twin of
make_golden.Tape: torch.randn / torch.randn_like hand out the supplied arrays in the reference's own consumption order and
check every requested shape.  The fixture holds outputs only: out32, the float64 truth as out32 + a float32 delta (the truth
to ~1e-13 relative), and one CRC-32 per member over the recipe's arrays.

    python tests/golden/make_golden_scale.py        # ~10 min on 8 cores, 1 MKL thread (fixed summation order)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import scale_recipe as R  # noqa: E402
from make_golden import import_reference, pretrained  # noqa: E402


Player = R.Player


def as_double(srm, m):
    md = srm.SWAGModel(dict(m.hparams)).init_params(dict(m.swa_params)).double()
    md.w_avg, md.w2_avg, md.pre_D = m.w_avg.double(), m.w2_avg.double(), m.pre_D.double()
    md.eval()
    return md


def main():
    srm = import_reference()
    torch.set_num_threads(1)
    M, J, B, NB = R.MEMBERS, R.DRAWS, R.SYSTEMS, R.NOISY_SYSTEMS
    out32 = np.empty((M, J, B, 2), np.float32)
    out64 = np.empty((M, J, B, 2), np.float64)
    noisy32 = np.empty((M, NB, 2), np.float32)
    noisy64 = np.empty((M, NB, 2), np.float64)
    t0 = time.time()
    for mi in range(M):
        m = srm.load_swag(pretrained(mi)).cpu()
        m.eval()
        md = as_double(srm, m)
        x = torch.tensor(R.x_block(mi))
        xd = x.double()
        for j in range(J):
            with Player(R.draw_noise(mi, j), torch.float32), torch.no_grad():
                out32[mi, j] = m.forward_swag_fast(x, scale=0.5).numpy()
            with Player(R.draw_noise(mi, j, dtype=np.float64), torch.float64), torch.no_grad():
                out64[mi, j] = md.forward_swag_fast(xd, scale=0.5).numpy()
        m.load(m.w_avg)
        md.load(md.w_avg)
        with Player(R.noisy_noise(mi), torch.float32), torch.no_grad():
            noisy32[mi] = m.forward(x[:NB], noisy_val=True).numpy()
        with Player(R.noisy_noise(mi, dtype=np.float64), torch.float64), torch.no_grad():
            noisy64[mi] = md.forward(xd[:NB], noisy_val=True).numpy()
        rel = np.abs(out32[mi] - out64[mi]) / np.abs(out64[mi])
        print(f"member {mi:2d}: reference fp32 vs its own float64: max rel {rel.max():.2e}, beyond 1e-5: {(rel > 1e-5).sum()}"
              f"  [{time.time() - t0:.0f} s]", flush=True)
    path = os.path.join(HERE, "case_scale.npz")
    np.savez(path, out32=out32, truth_delta=(out64 - out32.astype(np.float64)).astype(np.float32),
             noisy32=noisy32, noisy_truth_delta=(noisy64 - noisy32.astype(np.float64)).astype(np.float32),
             crc=R.checksums(), shape=np.array([M, J, B, NB]), scale=np.array(0.5))
    print(f"wrote case_scale.npz: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
