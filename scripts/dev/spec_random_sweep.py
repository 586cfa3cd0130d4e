#!/usr/bin/env python3
"""One-off validation (GPU): seeded random networks -- widths 1..128, depths 0..3, 41 / 82 features, fix_megno, random column masks,
random T -- through EVERY candidate form of the specialiser (waves x variant), each against the ahead-of-time generic engine bit for
bit (quiet and noisy).  usage: python scripts/dev/spec_random_sweep.py [trials [seed]]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bnn_chaos_model_amd import _native as N, ops, specialize as S  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
nforms = 0
for trial in range(trials):
    F = 82 if trial % 5 == 4 else 41
    H = int(rng.choice([1, 3, 8, 17, 24, 40, 48, 49, 64, 77, 96, 100, 128]))
    L = int(rng.choice([1, 2, 5, 12, 16, 20, 31, 48, 63]))
    din, dout = int(rng.integers(0, 4)), int(rng.integers(0, 3))
    megno = bool(rng.integers(0, 2))
    mask = int(rng.integers(0, 1 << 41)) | ((1 << 7) if megno else 0)
    if trial % 4 == 1:
        mask = ops.V50_ZERO_MASK | ((1 << 7) if megno else 0)
    T = int(rng.choice([2, 3, 5, 8, 37, 100]))
    try:
        plan = N.Plan(mask, 0.5, n_features=F, hidden=H, latent=L, fix_megno=megno, depth_in=din, depth_out=dout)
    except N.NativeError as e:
        print(trial, (F, H, L, din, dout, megno), "unsupported:", str(e)[:80])
        continue
    B = 53
    x = dev((rng.standard_normal((B, 1, F)) + 0.2 * rng.standard_normal((B, T, F))).astype(np.float32))
    W = dev((rng.standard_normal((3, plan.d)) * (0.6 / np.sqrt(max(H, 8)))).astype(np.float32))
    t0 = time.time()
    for nz in (False, True):
        kw = dict(philox_seed=9, draw_id0=3, system_id0=11, plan=plan, noisy=nz, debug=True)
        ref = ops.forward(x, W, engine="generic", **kw)
        try:
            cands = S.candidates(plan.arch, nz)
        except N.NativeError as e:
            print(trial, "no candidate:", str(e)[:80])
            continue
        for image, info in cands:
            plan.attach_spec(image, nz, info["w8"], info["flags"])
            got = ops.forward(x, W, engine="spec", **kw)
            ok = all(torch.equal(u, v) for u, v in zip(ref, got))
            nforms += 1
            if not ok:
                print("MISMATCH", trial, (F, H, L, din, dout, megno, hex(mask), T), nz, info)
                sys.exit(1)
    print(trial, (F, H, L, din, dout, megno, T), "ok, %d candidates, %.0f s" % (len(cands), time.time() - t0), flush=True)
print("all equal:", nforms, "forms")
