#!/usr/bin/env python3
"""One GPU's share of BASELINE configs[3] and configs[4], at full size, for the record.

  c4: 10M systems x 3000 draws over 8 GPUs -> 1.25M systems (20.5 GB of x) per GPU, draws in slabs of 250 reduced to
      float64 moments (MultiSwagSharded.local_moments): samples never exceed 2.5 GB.
  c4q: the same share, but what the scripts consume instead of moments: statistics epilogue fused in the forward tail, per-system
      quantile sketch (MultiSwagSharded.local_bands): median / 68 % / 95 % bands + mean of the post-epilogue times.
  c5: x [1e6, 3, 100, 41] over 8 GPUs -> 375k rows (6.15 GB) per GPU, 10 chunks x 100 samples: one (seed, draw) per chunk
      per sample, samples [100, 375k, 2] kept (the scripts consume them).
  c5q: the same share streamed: FeatureRegressor-style bands per 5-planet system (min over its 3 trios), nothing kept.
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from bnn_chaos_model_amd import ops  # noqa: E402
from bnn_chaos_model_amd.distributed import MultiSwagSharded  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda")
wa, w2, pd = bench.synthetic_ensemble(30, dev)


def big_x(B, piece=125_000):
    x = torch.empty((B, 100, 41), dtype=torch.float32, device=dev)
    for i, lo in enumerate(range(0, B, piece)):
        hi = min(B, lo + piece)
        x[lo:hi] = bench.synthetic_x(hi - lo, dev, 1000 + i)
    return x


if which == "c4":
    B, J = 1_250_000, 3000
    x = big_x(B)
    idx = (torch.arange(J, dtype=torch.int32) % 30).to(dev)
    drv = MultiSwagSharded(wa, w2, pd, draws_per_launch=250)
    drv.local_moments(x[:10_000], idx[:250], 7, 0)  # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mom = drv.local_moments(x, idx, philox_seed=7, system_id0=0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert torch.isfinite(mom).all()
    print(f"c4 share: {B} systems x {J} draws = {B * J:.3g} evals in {dt:.2f} s = {B * J / dt:.4g} evals/s; "
          f"peak GPU memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
elif which == "c4q":
    B, J = 1_250_000, 3000
    x = big_x(B)
    idx = (torch.arange(J, dtype=torch.int32) % 30).to(dev)
    drv = MultiSwagSharded(wa, w2, pd, draws_per_launch=250)
    q = (2.5, 16.0, 50.0, 84.0, 97.5)
    drv.local_bands(x[:10_000], idx[:250], q, 7, 0)  # warm-up
    torch.cuda.reset_peak_memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bands = drv.local_bands(x, idx, q, philox_seed=7, system_id0=0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert bands.shape == (B, 6) and torch.isfinite(bands).all()
    print(f"c4q share: {B} systems x {J} draws = {B * J:.3g} evals -> bands [B,5] + mean in {dt:.2f} s = {B * J / dt:.4g} evals/s; "
          f"peak GPU memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB (x = {x.numel() * 4 / 2**30:.1f} GiB, sketch = "
          f"{946 * 4 * B / 2**30:.1f} GiB)", flush=True)
elif which == "c5q":
    import numpy as np
    B, chunks, samples, trios = 375_000, 10, 100, 3
    x = big_x(B)
    sk = ops.QuantileSketch(B, group=trios)
    st = ops.stats_params()
    rng = np.random.default_rng(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s0 in range(0, samples, 25):
        idx = torch.as_tensor(rng.integers(0, 30, 25 * chunks).astype(np.int32))
        sk.update(ops.multiswag_stats(x, wa, w2, pd, idx, st=st, nchunks=chunks, philox_seed=3, draw_id0=s0 * chunks))
    bands = sk.percentiles((2.5, 16.0, 50.0, 84.0, 97.5))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert bands.shape == (B // trios, 5) and torch.isfinite(bands).all()
    print(f"c5q share: {B} rows x {samples} samples ({chunks} chunks) = {B * samples:.3g} evals -> bands per 5-planet system in "
          f"{dt * 1e3:.1f} ms = {B * samples / dt:.4g} evals/s; peak GPU memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
else:
    B, chunks, samples = 375_000, 10, 100
    x = big_x(B)
    J = chunks * samples
    idx = torch.randint(0, 30, (J,), dtype=torch.int32, device=dev)
    ops.multiswag(x[:1000], wa, w2, pd, idx[:10], nchunks=10, philox_seed=3)  # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ops.multiswag(x, wa, w2, pd, idx, nchunks=chunks, philox_seed=3)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert out.shape == (samples, B, 2) and torch.isfinite(out).all()
    print(f"c5 share: {B} rows x {samples} samples ({chunks} chunks, {J} draws) = {B * samples:.3g} evals in {dt * 1e3:.1f} ms = "
          f"{B * samples / dt:.4g} evals/s; peak GPU memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
