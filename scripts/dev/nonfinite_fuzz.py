#!/usr/bin/env python3
"""One-off validation (GPU): the random-damage comparison of tests/test_hip_nonfinite.py::test_random_damage_against_the_oracle over many
seeds and heavier damage (up to 8 damaged elements per system, whole rows / columns, the dead column), every network of that test, quiet
and noisy, explicit and in-kernel noise: NaN pattern identical to the oracle's, 1e-5 (2e-5 noisy) where finite.
usage: python scripts/dev/nonfinite_fuzz.py [seeds]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bnn_chaos_model_amd import ops  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import test_hip_nonfinite as T  # noqa: E402

z = T.load_golden("case_nonfinite.npz")
dev = T.dev
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
checked = nan_sys = fin_dmg = 0
for seed in range(nseeds):
    rng = np.random.default_rng(1000 + seed)
    for net in ("v50", "dead", "h48megno", "deriv82", "lin0out8"):
        if net == "v50":
            plan, arch, w = ops.get_plan(), orc.make_arch(T=100), z["v50_0_swagfast_w"]
            base = np.tile(T.load_golden("inputs.npz")["x_slow"], (4, 1, 1))
        elif net == "dead":
            (plan, arch), w = T.dead_plan(ops, orc, z), z["dead_swagfast_w"]
            base = np.tile(T.load_golden("inputs.npz")["x_slow"], (4, 1, 1))
        else:
            plan, arch, w, xa = T._arch_case(ops, orc, net)
            base = np.tile(xa, (128 // xa.shape[0] + 1, 1, 1))
        B = 128
        base = base[:B].copy() + 0.01 * rng.standard_normal(base[:B].shape).astype(np.float32)
        NF = base.shape[2]
        x = base.copy()
        vals = (np.nan, np.inf, -np.inf)
        for b in range(B):
            kind = int(rng.integers(0, 6))
            if kind == 0:
                continue
            if kind <= 3:
                for _ in range(int(rng.integers(1, 9))):
                    x[b, int(rng.integers(0, 100)), int(rng.integers(0, NF))] = vals[int(rng.integers(0, 3))]
            elif kind == 4:
                x[b, int(rng.integers(0, 100)), :] = vals[int(rng.integers(0, 3))]
            else:
                x[b, :, int(rng.integers(0, NF))] = vals[int(rng.integers(0, 3))]
        if net == "dead":
            col = int(z["dead_col"])
            for b in range(0, B, 5):
                x[b] = base[b]
                x[b, int(rng.integers(0, 100)), col] = np.inf
        L, SM = arch.latent, 2 * arch.latent + 2 * int(arch.fix_megno)
        W = dev(w[None])
        for noisy in (False, True):
            for explicit in (True, False):
                if explicit:
                    e1, e2 = rng.standard_normal((B, L), dtype=np.float32), rng.standard_normal((B, L), dtype=np.float32)
                    kw, okw = dict(eps=dev(np.stack([e1, e2], 1)[None])), {}
                    if noisy:
                        e_in, e_sum = rng.standard_normal((B, 100, NF), dtype=np.float32), rng.standard_normal((B, SM), dtype=np.float32)
                        kw.update(eps_in=dev(e_in[None]), eps_sum=dev(e_sum[None]))
                        okw = dict(eps_in=e_in, eps_sum=e_sum)
                else:
                    kw = dict(philox_seed=seed, draw_id0=2, system_id0=77, noisy=noisy)
                    eps = ops.philox_normal(2, seed, 2, 1, B=B, system_id0=77, width=L).cpu().numpy()[0]
                    e1, e2, okw = eps[:, 0], eps[:, 1], {}
                    if noisy:
                        okw = dict(eps_in=ops.philox_normal(3, seed, 2, 1, B=B, system_id0=77, width=100, n_features=NF).cpu().numpy()[0],
                                   eps_sum=ops.philox_normal(4, seed, 2, 1, B=B, system_id0=77, width=SM).cpu().numpy()[0])
                got = ops.forward(dev(x), W, plan=plan, **kw)[0].cpu().numpy()
                want = orc.forward(x, w, e1, e2, arch=arch, **okw)
                T.same_nan_close_elsewhere(got, want, rtol=2e-5 if noisy else 1e-5)
                dmg = ~np.isfinite(x).all(axis=(1, 2))
                checked += B
                nan_sys += int(np.isnan(got).any(1).sum())
                fin_dmg += int((dmg & np.isfinite(got).all(1)).sum())
    print(f"seed {seed}: ok", flush=True)
print(json.dumps({"evaluations_checked": checked, "nan_systems": nan_sys, "damaged_but_finite": fin_dmg, "seeds": nseeds}))
