"""Closed-form inputs and noise for the parity-at-scale fixture (case_scale.npz).

Everything here is INTEGER arithmetic on uint64 counters (a SplitMix64 finaliser) followed by ONE correctly rounded IEEE
operation per value, so numpy reproduces every array bit for bit on any machine: no torch / numpy generator, no libm call.
The fixture therefore stores only the reference's OUTPUTS (and CRC-32s of what this module makes); the generator
(make_golden_scale.py, build container, imports the unmodified reference) and the tests (any box) both call this module.

A "normal" is an Irwin-Hall sum of 16-bit fields of the hash: 12 fields (three hashes, variance (2^32 - 1) / 2^32, support
+-6) for every noise tensor the reference draws, 4 fields (one hash, support +-3.46 sigma) for the bulk of x.  Parity does
not care about the tails; what matters is that both sides see the same float32 numbers.

x follows SURVEY.md section 8(d)'s "slow" distribution: x[b,t,f] = base[b,f] + 0.1 n[b,t,f], column 0 = the standardised
time ramp from -1.71 to 1.74 (spock_reg_model.py:934, 945) -- mu spreads over the whole (4, 12) range.
"""
import zlib

import numpy as np

T, F, D, K, LATENT = 100, 41, 7583, 30, 20
MEMBERS, DRAWS, SYSTEMS = 30, 2, 4096        # forward_swag_fast: every member x 2 draws x its own block of 4096 systems
NOISY_SYSTEMS = 512                           # VarModel.forward(noisy_val=True) at w_avg: the first 512 systems of each block

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)
(S_BASE, S_N, S_Z1, S_Z2, S_EPS, S_EPS_IN, S_EPS_SUM, S_EPS_NOISY) = range(1, 9)


def _mix(z):
    """SplitMix64 finaliser, wrapping uint64 arithmetic (numpy arrays wrap silently); works in place on a fresh array."""
    z += _G
    t = z >> np.uint64(30)
    z ^= t
    z *= _M1
    np.right_shift(z, np.uint64(27), out=t)
    z ^= t
    z *= _M2
    np.right_shift(z, np.uint64(31), out=t)
    z ^= t
    return z


def _fields_sum(h):
    m = np.uint64(0xFFFF)
    s = h & m
    t = h >> np.uint64(16)
    t &= m
    s += t
    np.right_shift(h, np.uint64(32), out=t)
    t &= m
    s += t
    np.right_shift(h, np.uint64(48), out=t)
    s += t
    return s.view(np.int64)


def _counter(stream, idx):
    return (np.uint64(stream) << np.uint64(58)) | (idx.astype(np.uint64) << np.uint64(2))


def isum4(stream, idx):
    """Centred sum of the four 16-bit fields of one hash: int64 in [-131070, 131070], variance 1431655765."""
    return _fields_sum(_mix(_counter(stream, idx))) - 131070


def normal12(stream, idx, dtype=np.float32):
    """Twelve fields / 65536: exactly representable in float32 (|numerator| < 2^19, power-of-two divisor)."""
    c = _counter(stream, idx)
    s = _fields_sum(_mix(c.copy())) + _fields_sum(_mix(c | np.uint64(1))) + _fields_sum(_mix(c | np.uint64(2))) - 393210
    return (s.astype(np.float64) / 65536.0).astype(dtype)


_XCHUNK = 128     # systems per pass: the uint64 temporaries (4 MB each) stay in cache


def x_block(member, b0=0, nb=SYSTEMS):
    """Rows [b0, b0 + nb) of member's block of systems -> float32 [nb, T, F]."""
    f = np.arange(F, dtype=np.int64)
    t = np.arange(T, dtype=np.int64)
    ramp = ((345 * t - 171 * 99).astype(np.float64) / 9900.0).astype(np.float32)
    x = np.empty((nb, T, F), np.float32)
    for c0 in range(0, nb, _XCHUNK):
        sysid = member * SYSTEMS + b0 + np.arange(c0, min(c0 + _XCHUNK, nb), dtype=np.int64)
        base = isum4(S_BASE, sysid[:, None] * F + f[None, :])                                   # [n, F]
        n = isum4(S_N, (sysid[:, None, None] * T + t[None, :, None]) * F + f[None, None, :])   # [n, T, F]
        n += 10 * base[:, None, :]
        x[c0:c0 + len(sysid)] = n.astype(np.float64) / 378370.0                                 # sd of isum4 = 37837.2; ONE rounding to fp32
    x[:, :, 0] = ramp[None, :]
    return x


def draw_noise(member, draw, nb=SYSTEMS, dtype=np.float32, d=D, k=K, latent=LATENT):
    """What sample_weights + compute_summary_stats consume in one forward_swag_fast call:
    z1 [1, d], z2 [k, 1], eps1 [nb, latent], eps2 [nb, latent].  (d, k, latent: the network's; the defaults are the pretrained ensemble's.
    Strides of the counters use the LARGEST sizes any fixture asks for, so streams of different networks never overlap within a member.)"""
    j = member * DRAWS + draw
    DS, KS, LS = max(d, 65536), max(k, 64), max(latent, 128)
    if (d, k, latent) == (D, K, LATENT):
        DS, KS, LS = D, K, LATENT          # (the pretrained fixture's counters, as generated in round 6)
    z1 = normal12(S_Z1, j * DS + np.arange(d, dtype=np.int64), dtype)[None, :]
    z2 = normal12(S_Z2, j * KS + np.arange(k, dtype=np.int64), dtype)[:, None]
    b = np.arange(nb, dtype=np.int64)[:, None]
    l = np.arange(latent, dtype=np.int64)[None, :]
    e = [normal12(S_EPS, ((j * SYSTEMS + b) * 2 + kind) * LS + l, dtype) for kind in (0, 1)]
    return z1, z2, e[0], e[1]


def noisy_noise(member, nb=NOISY_SYSTEMS, dtype=np.float32, latent=LATENT, summary=None):
    """What VarModel.forward(noisy_val=True) consumes: randn_like(x) [nb,T,F], eps1, eps2 [nb,latent], randn_like(summary) [nb,summary]
    (summary = 2 * latent, + 2 with fix_megno)."""
    summary = 2 * latent if summary is None else summary
    LS = LATENT if (latent, summary) == (LATENT, 2 * LATENT) else 128
    b = np.arange(nb, dtype=np.int64)
    e_in = normal12(S_EPS_IN, ((member * SYSTEMS + b)[:, None, None] * T + np.arange(T, dtype=np.int64)[None, :, None]) * F
                    + np.arange(F, dtype=np.int64)[None, None, :], dtype)
    l = np.arange(latent, dtype=np.int64)[None, :]
    e = [normal12(S_EPS_NOISY, ((member * SYSTEMS + b[:, None]) * 2 + kind) * LS + l, dtype) for kind in (0, 1)]
    s = np.arange(summary, dtype=np.int64)[None, :]
    e_sum = normal12(S_EPS_SUM, (member * SYSTEMS + b[:, None]) * 2 * LS + s, dtype)
    return e_in, e[0], e[1], e_sum


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def checksums(members=range(MEMBERS)):
    """One CRC-32 per member over its x block and every noise array: stored in the fixture, re-derived by the tests."""
    out = []
    for m in members:
        c = crc(x_block(m))
        for j in range(DRAWS):
            for a in draw_noise(m, j):
                c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
        for a in noisy_noise(m):
            c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
        out.append(c & 0xFFFFFFFF)
    return np.array(out, np.uint32)


class Player:
    """The playing twin of make_golden.Tape: torch.randn / torch.randn_like hand out the queued arrays in order, and a request
    whose shape is not the next queued one is an error.  Used around the UNMODIFIED reference by make_golden_scale.py and around
    this repository's module surface by the GPU test: both sides consume the same numbers in the reference's order."""

    def __init__(self, arrays, dtype=None):
        import torch
        self.torch = torch
        self.queue = [torch.tensor(np.ascontiguousarray(a), dtype=dtype or torch.float32) for a in arrays]
        self._orig = None

    def _next(self, shape, device=None):
        assert self.queue, "more normals were drawn than were queued"
        t = self.queue.pop(0)
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t if device is None else t.to(device)

    def __enter__(self):
        torch = self.torch
        self._orig = (torch.randn, torch.randn_like)

        def randn(*size, **k):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            return self._next(size, k.get("device"))

        def randn_like(t, **k):
            return self._next(t.shape, k.get("device", t.device))

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        self.torch.randn, self.torch.randn_like = self._orig
        if exc[0] is None:
            assert not self.queue, f"{len(self.queue)} queued arrays were never drawn"
