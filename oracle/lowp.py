"""numpy emulation of the reduced-precision feature_nn (TEST INFRASTRUCTURE, like the rest of oracle/): what
bnn_chaos_model_amd/csrc/bnn_lowp.hip.h computes, up to the matrix pipe's fp32 accumulation order.

Operands (x incl. the constant 1.0 of the bias slot, weights, biases, post-ReLU activations) are split into `ns` bfloat16 (or IEEE half) parts
(round to nearest even; part p = bf16 of what is left after parts < p), the products of order <= ns - 1 are summed (here in
float64; on the GPU in fp32 inside v_mfma_f32_16x16x32_bf16), ReLU, re-split.  Reference network: spock_reg_model.py:301-321,
359, 417 with the v50 column mask (:452-500)."""
import numpy as np

OFF = dict(W1=81, B1=1721, W2=1761, B2=3361, W3=3401, B3=4201)
LIVE = [0] + list(range(8, 38))


def to_bf16(a):
    """Round-to-nearest-even to bfloat16, returned as float32 (finite inputs)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def to_f16(a):
    """Round-to-nearest-even to IEEE half, returned as float32."""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def split(a, ns, fmt="bf16"):
    parts, rest = [], np.asarray(a, np.float32)
    rnd = to_f16 if fmt == "f16" else to_bf16
    for _ in range(ns):
        p = rnd(rest)
        parts.append(p)
        rest = (rest - p).astype(np.float32)   # exact in fp32
    return parts


def split_matmul(act, W, b, ns, fmt="bf16"):
    """act [..., K] fp32, W [N, K], b [N] -> [..., N] float64: sum over part products of order <= ns - 1, bias likewise
    (its activation is the constant 1.0, whose parts are (1, 0, 0))."""
    ap, wp, bp = split(act, ns, fmt), split(W, ns, fmt), split(b, ns, fmt)
    out = 0.0
    for order in range(ns):
        for i in range(order + 1):
            out = out + ap[i].astype(np.float64) @ wp[order - i].astype(np.float64).T
    for p in bp:
        out = out + p.astype(np.float64)
    return out


def feature_nn(x, w, ns, fmt="bf16"):
    """x [B,T,41] fp32 (columns outside the v50 mask are ignored), w [7583] -> latents [B,T,20] float64."""
    W1 = w[OFF["W1"]:OFF["B1"]].reshape(40, 41)[:, LIVE]
    W2 = w[OFF["W2"]:OFF["B2"]].reshape(40, 40)
    W3 = w[OFF["W3"]:OFF["B3"]].reshape(20, 40)
    b1, b2, b3 = w[OFF["B1"]:OFF["B1"] + 40], w[OFF["B2"]:OFF["B2"] + 40], w[OFF["B3"]:OFF["B3"] + 20]
    h = np.maximum(split_matmul(x[..., LIVE], W1, b1, ns, fmt), 0).astype(np.float32)
    h = np.maximum(split_matmul(h, W2, b2, ns, fmt), 0).astype(np.float32)
    return split_matmul(h, W3, b3, ns, fmt)


def pooled_summary(lat):
    """compute_summary_stats (:418-431) with both noise draws = 0: [mean over time | sqrt(unbiased var + 1e-5)] in float64."""
    mu = lat.mean(1)
    var = lat.var(1, ddof=1)
    return np.concatenate([mu, np.sqrt(np.abs(var) + 1e-5)], 1)
