"""Post-sampling statistics of the evaluation scripts on the GPU.

    fast_truncnorm   figures/multiswag_5_planet.py:306-370 (= figures/main_figures.py:167-227)
    resample_prior   figures/multiswag_5_planet.py:396-422
    min_over_trios   figures/multiswag_5_planet.py:428
    percentiles      :484-489 and np.median of figures/main_figures.py:277-278  (ops.quantiles)

rng="numpy" consumes numpy's global generator exactly as the reference does (np.random.normal(size=(nsamp, cd)) per chunk
of d elements; np.random.rand(n_samples)) and feeds the draws to the kernels, so np.random.seed(s) reproduces the
reference's numbers bit for bit; rng="philox" generates the noise in-kernel.
"""

import numpy as np
import torch

from . import _native as N
from . import ops
from .spock_reg_model import _gpu

_NORMALIZATION = None


def prior_pdf(logT):
    """The unnormalised prior of :400-403."""
    return 3.27086190404742 * np.exp(-0.424033970670719 * logT) - 10.8793430454878 * np.exp(-0.200351029031774 * logT ** 2)


def prior_normalization():
    global _NORMALIZATION
    if _NORMALIZATION is None:
        from scipy.integrate import quad
        _NORMALIZATION = quad(prior_pdf, a=9, b=np.inf)[0]  # :404
    return _NORMALIZATION


def fast_truncnorm(loc, scale=None, left=np.inf, right=np.inf, d=10000, nsamp=50, seed=0, rng="numpy"):
    """First of `nsamp` Gaussian candidates inside (left, right) (else the first candidate), elementwise, with the reference's
    acceptance test (:352-358: right = inf -> v > left; left = inf -> v < right; else both); returns a tensor shaped like
    `scale` on the GPU.  `loc` may also be the [..., 2] (mu, std) tensor of the forward, with scale=None."""
    g = _gpu()
    if scale is None:
        musd = torch.as_tensor(loc).to(g, torch.float32).contiguous()
        shape = musd.shape[:-1]
    else:
        loc_t, scale_t = torch.as_tensor(loc).to(g, torch.float32), torch.as_tensor(scale).to(g, torch.float32)
        shape = scale_t.shape
        musd = torch.stack([loc_t.reshape(-1), scale_t.reshape(-1)], dim=1).contiguous()
    n = int(np.prod(shape)) if len(shape) else 1
    out = torch.empty(n, dtype=torch.float32, device=g)
    normals = None
    if rng == "numpy":
        host = np.empty((nsamp, n), np.float64)
        for start in range(0, n, d):            # :333-345: one (nsamp, cd) draw per chunk of d elements
            end = min(start + d, n)
            host[:, start:end] = np.random.normal(size=(nsamp, end - start))
        normals = torch.as_tensor(host).to(g)
    elif rng != "philox":
        raise ValueError("rng must be 'numpy' or 'philox'")
    N.check(N.lib().bnn_truncnorm_f32(N.ptr(musd), n, N.ptr(normals), int(nsamp), float(left), float(right), int(seed), 0, N.ptr(out), N.stream_ptr()))
    return out.reshape(tuple(shape))


def resample_prior(samps, threshold=9.0, rng="numpy", seed=0):
    """samps[samps >= 9] = inv_cdf(uniform), in C order (:396-422) -> new tensor on the GPU."""
    g = _gpu()
    vals = torch.as_tensor(samps).to(g, torch.float32).contiguous().clone()
    flat = vals.view(-1)
    mask = flat >= threshold
    n_samples = int(mask.sum().item())  # the table size depends on it (:412-413): one host sync, as in the reference
    if n_samples == 0:
        return vals
    rank = (torch.cumsum(mask, 0) - 1).to(torch.int64).contiguous()
    bins = n_samples * 4
    top = 100.0
    edges = np.linspace(9, top, num=bins)
    cum = np.array([0] + list(np.cumsum(prior_pdf(edges) / prior_normalization() * (edges[1] - edges[0]))) + [1], dtype=np.float64)
    edges = np.array([9.0] + list(edges) + [top], dtype=np.float64)
    order = np.argsort(cum, kind="mergesort")  # interp1d(assume_sorted=False) sorts its x (:418)
    cum_d, edge_d = torch.as_tensor(cum[order]).to(g), torch.as_tensor(edges[order]).to(g)
    u = None
    if rng == "numpy":
        u = torch.as_tensor(np.random.rand(n_samples)).to(g)  # :419
    elif rng != "philox":
        raise ValueError("rng must be 'numpy' or 'philox'")
    N.check(N.lib().bnn_prior_resample_f32(N.ptr(flat), flat.numel(), N.ptr(rank), N.ptr(cum_d), N.ptr(edge_d), cum_d.numel(), N.ptr(u),
                                           float(threshold), int(seed), 0, N.stream_ptr()))
    return vals


def min_over_trios(samps):
    """np.min(samps_time, 2).T (:428): [samples, sims, trios] -> [sims, samples]."""
    g = _gpu()
    v = torch.as_tensor(samps).to(g, torch.float32).contiguous()
    out = torch.empty(v.shape[:-1], dtype=torch.float32, device=g)
    N.check(N.lib().bnn_group_min_f32(N.ptr(v), out.numel(), v.shape[-1], N.ptr(out), N.stream_ptr()))
    return out.T


def percentiles(outs, q):
    """np.percentile(outs[i], q) for every simulation i: outs [sims, samples] -> [sims, len(q)] (:484-489)."""
    o = torch.as_tensor(outs).to(_gpu(), torch.float32)
    s = torch.stack([o.T, o.T], dim=2).contiguous()  # [samples, sims, 2]: reuse the two-channel kernel
    return ops.quantiles(s, q)[:, 0, :]
