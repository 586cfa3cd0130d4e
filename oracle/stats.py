"""numpy restatement of the post-sampling statistics (TEST INFRASTRUCTURE, like the rest of oracle/).

figures/multiswag_5_planet.py: fast_truncnorm :306-370 (same function in figures/main_figures.py:167-227), prior
resampling of samples past 9 :396-422, min over trios :428.  Pinned by tests/golden/case_stats.npz, which
make_golden.py produces by executing the reference's own source fragments."""
import numpy as np


def truncnorm_first_good(loc, scale, normals, left, right=np.inf):
    """fast_truncnorm: candidates = normals * scale + loc (float64), first one inside the interval (:352-358), else the first.
    loc, scale: any shape (float32); normals [nsamp, n] float64 in flattened element order -> samples like scale."""
    sc = np.asarray(scale).reshape(-1)
    lc = np.asarray(loc).reshape(-1)
    rand_out = normals * sc[None] + lc[None]                      # :347-350
    if right == np.inf:                                           # :352-358
        mask = rand_out > left
    elif left == np.inf:
        mask = rand_out < right
    else:
        mask = (rand_out > left) & (rand_out < right)
    first_good = rand_out[mask.argmax(0), np.arange(sc.size)]     # :360-362
    out = np.zeros_like(sc)                                       # float32 like `scale` (:329)
    out[:] = first_good
    return out.reshape(np.asarray(scale).shape)


def prior_pdf(logT):
    return 3.27086190404742 * np.exp(-0.424033970670719 * logT) - 10.8793430454878 * np.exp(-0.200351029031774 * logT ** 2)  # :400-403


def prior_table(n_samples, normalization, top=100.0):
    """cum_values, bin_edges of :413-417."""
    bins = n_samples * 4
    edges = np.linspace(9, top, num=bins)
    cum = [0] + list(np.cumsum(prior_pdf(edges) / normalization * (edges[1] - edges[0]))) + [1]
    return np.array(cum, dtype=np.float64), np.array([9.0] + list(edges) + [top], dtype=np.float64)


def interp1d_linear(x, y, xn):
    """scipy.interpolate.interp1d(x, y)(xn), kind='linear', assume_sorted=False (:418, :420)."""
    order = np.argsort(x, kind="mergesort")
    x, y = x[order], y[order]
    idx = np.searchsorted(x, xn).clip(1, len(x) - 1)
    lo, hi = idx - 1, idx
    slope = (y[hi] - y[lo]) / (x[hi] - x[lo])
    return slope * (xn - x[lo]) + y[lo]


def resample_prior(samps, u, normalization, threshold=9.0):
    """samps[samps >= 9] = inv_cdf(u), in C order (:396, :419-422)."""
    out = samps.copy()
    mask = out >= threshold
    n = int(mask.sum())
    if n:
        cum, edges = prior_table(n, normalization)
        out[mask] = interp1d_linear(cum, edges, u[:n])
    return out


# ---- the streaming (Philox) form of the epilogue: restatement of bnn_chaos_model_amd/csrc/bnn_stats.hip.h -------------------
def prior_survival_table(thr=9.0, top=100.0, m=8192):
    """S(t_i) = P(T > t_i | T >= thr) of the prior (:400-404) in closed form (float64), rounded to fp32; returns (S, step)."""
    from scipy.special import erfc
    a, b, c, d = 3.27086190404742, 0.424033970670719, 10.8793430454878, 0.200351029031774
    G = lambda t: a / b * np.exp(-b * t) - c * 0.5 * np.sqrt(np.pi / d) * erfc(np.sqrt(d) * t)
    step = (top - thr) / (m - 1)
    S = (G(thr + step * np.arange(m)) / G(thr)).astype(np.float32)
    S[0] = 1.0
    return S, step


def stream_epilogue(musd, cand, level, left=4.0, thr=9.0, top=100.0, m=8192):
    """musd [R,B,2] fp32; cand [R,B,nsamp] fp32 normals; level [R,B] fp32 in (0,1] -> t [R,B] fp32, in fp32 arithmetic:
    first candidate z*sd+mu above `left` (else the first one); values >= thr are replaced by the prior's inverse survival
    function at `level` (bisection on the fp32 table + linear interpolation)."""
    mu, sd = musd[..., 0].astype(np.float32), musd[..., 1].astype(np.float32)
    v = (cand * sd[..., None]).astype(np.float32) + mu[..., None]            # fp32 multiply, then fp32 add (no fma)
    ok = v > np.float32(left)
    first = np.take_along_axis(v, ok.argmax(-1)[..., None], -1)[..., 0]
    t = first.astype(np.float32)
    S, step = prior_survival_table(thr, top, m)
    step = np.float32(step)
    sel = t >= np.float32(thr)
    lv = level[sel].astype(np.float32)
    # lo = last knot with S[lo] >= v  (S is non-increasing): searchsorted on the reversed (ascending) table
    lo = (m - np.searchsorted(S[::-1], lv, side="left") - 1).clip(0, m - 2)
    hi = lo + 1
    a, b = S[lo], S[hi]
    frac = ((a - lv) / (a - b)).astype(np.float32)
    val = np.float32(thr) + step * (lo.astype(np.float32) + frac)
    tail = ~(lv > S[m - 1])
    val = np.where(tail, np.float32(thr) + step * np.float32(m - 1), val).astype(np.float32)
    t = t.copy()
    t[sel] = val
    return t
