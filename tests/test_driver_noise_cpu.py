"""The parity-mode MC driver draws the reference's generators in the reference's order -- straight into preallocated rows
(regression.draw_reference_noise: `torch.randn(..., out=view)`).  Pinned here, without a GPU, against the tapes the unmodified
reference left in the fixtures: the 5-planet chunk loop (figures/multiswag_5_planet.py:295-298) and the 3-call MultiSWAG grid
(figures/spock/regression.py:74-92)."""
import time

import numpy as np
import torch

from conftest import load_golden, tape
from bnn_chaos_model_amd.regression import draw_reference_noise


def _check_against_tape(z, samples, B, chunks, seed):
    tp = tape(z)
    parts = torch.chunk(torch.arange(B), chunks)
    np.random.seed(seed)
    torch.manual_seed(seed)
    seed_idx, z1, z2, eps = draw_reference_noise(samples, parts, 2, 7583, 30, 20, torch.device("cpu"), torch.device("cpu"))
    nch = len(parts)
    assert len(tp) == 5 * samples * nch
    lo = 0
    for e in range(samples * nch):
        s_, c_ = divmod(e, nch)
        if c_ == 0:
            lo = 0
        kinds = [k for k, _ in tp[5 * e: 5 * e + 5]]
        assert kinds == ["np.randint", "torch.randn", "torch.randn", "torch.randn_like", "torch.randn_like"]
        assert int(tp[5 * e][1]) == seed_idx[e]
        assert np.array_equal(tp[5 * e + 1][1].reshape(-1), z1[e].numpy())
        assert np.array_equal(tp[5 * e + 2][1].reshape(-1), z2[e].numpy())
        n = len(parts[c_])
        assert np.array_equal(tp[5 * e + 3][1], eps[s_, 0, lo:lo + n].numpy())
        assert np.array_equal(tp[5 * e + 4][1], eps[s_, 1, lo:lo + n].numpy())
        lo += n
    # and the generators are left where the reference's loop leaves them
    return torch.randn(3), np.random.rand(3)


def test_chunk_loop_tape_is_reproduced_draw_for_draw():
    z = load_golden("case_chunk_loop.npz")
    _check_against_tape(z, int(z["samples"]), int(z["nrows"]), int(z["chunks"]), 5000)


def test_multiswag_grid_tape_is_reproduced_draw_for_draw(inputs):
    z = load_golden("case_multiswag_grid.npz")
    _check_against_tape(z, 3, inputs["slow"].shape[0], 1, 4000)


def test_generators_end_where_a_fresh_tensor_per_draw_leaves_them():
    """`out=` into a view consumes exactly what a fresh tensor of that shape does (small pools take the scalar path of normal_, large
    ones the vectorised one: both sizes here), and costs less host time per draw (reported, not asserted)."""
    S, d, K, L = 3, 7583, 30, 20
    for B, chunks in ((7, 3), (300, 10)):
        parts = torch.chunk(torch.arange(B), chunks)
        np.random.seed(1); torch.manual_seed(1)
        t0 = time.perf_counter()
        seed_idx, z1, z2, eps = draw_reference_noise(50, parts, S, d, K, L, torch.device("cpu"), torch.device("cpu"))
        t_new = time.perf_counter() - t0
        tail_new = (torch.randn(2), np.random.rand(2))
        np.random.seed(1); torch.manual_seed(1)
        t0 = time.perf_counter()
        ref = []
        for e in range(50 * len(parts)):
            n = len(parts[e % len(parts)])
            ref.append((np.random.randint(0, S), torch.randn((1, d))[0], torch.randn((K, 1))[:, 0], torch.randn(n, L), torch.randn(n, L)))
        t_old = time.perf_counter() - t0
        tail_old = (torch.randn(2), np.random.rand(2))
        assert torch.equal(tail_new[0], tail_old[0]) and np.array_equal(tail_new[1], tail_old[1])
        lo = 0
        for e, (si, a, b, e1, e2) in enumerate(ref):
            s_, c_ = divmod(e, len(parts))
            if c_ == 0:
                lo = 0
            n = e1.shape[0]
            assert si == seed_idx[e] and torch.equal(a, z1[e]) and torch.equal(b, z2[e])
            assert torch.equal(e1, eps[s_, 0, lo:lo + n]) and torch.equal(e2, eps[s_, 1, lo:lo + n])
            lo += n
        print(f"B={B} chunks={chunks}: {1e6 * t_new / len(ref):.1f} us per draw into rows, {1e6 * t_old / len(ref):.1f} us with a fresh tensor per draw")
