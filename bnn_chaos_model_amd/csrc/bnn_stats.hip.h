// bnn_stats.hip.h -- the post-sampling statistics of the evaluation scripts as ONE per-evaluation device routine, shared by the
// forward kernel's fused tail (bnn_multiswag_stats_f32) and by the stand-alone epilogue kernel (bnn_stats_draw_f32), so the two
// give the same bits:
//     (mu, std) -> fast_truncnorm(left = 4, nsamp = 40)          figures/multiswag_5_planet.py:306-370, 388-392
//               -> "Resample with prior" for values >= 9         :396-422
// All noise is Philox, keyed by (GLOBAL output row, GLOBAL system), so the result does not depend on sharding, slabs or
// launch mode.  (The numpy-replay forms of these steps, which consume the reference's generator draws, are the separate
// kernels bnn_truncnorm_f32 / bnn_prior_resample_f32.)
//
// Prior on [thr, top] (:400-404): p(t) ~ 3.27086190404742 exp(-0.424033970670719 t) - 10.8793430454878 exp(-0.200351029031774 t^2).
// The reference inverts a left-Riemann CDF table whose size depends on the number of samples to replace; here the table is the
// EXACT survival function S(t) = P(T > t | T >= thr) at m equally spaced knots (host, float64, closed form with erf), stored as
// fp32 (small values keep their relative precision, which a CDF near 1 would not), inverted by bisection + linear interpolation.
#pragma once
#include "bnn_common.hip.h"

namespace bnn {


// first of tn_nsamp candidates z * sd + mu above tn_left, else the first candidate (argmax of an all-False mask is 0, :360-362)
DEVINL float stats_truncnorm(const StatsParams& sp, float mu, float sd, int64_t row, int64_t sys, uint64_t seed) {
    float first = 0.0f, pick = 0.0f;
    bool found = false;
    for (int s0 = 0; s0 < sp.tn_nsamp && !found; s0 += 4) {
        const f32x4 z4 = philox_sys4(TAG_TNS, row, sys, s0 >> 2, seed);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (s0 + k < sp.tn_nsamp && !found) {
                const float v = z4[k] * sd + mu;  // rand_out * scale + loc (:347-350), fp32 here
                if (s0 + k == 0) first = v;
                if (v > sp.tn_left) { pick = v; found = true; }
            }
        }
    }
    return found ? pick : first;
}

DEVINL float stats_prior_draw(const StatsParams& sp, int64_t row, int64_t sys, uint64_t seed) {
    const uint4 c = philox_sys_ctr(TAG_US, row, sys, 0);
    const uint4 q = philox4x32_10(c, make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const float v = ((float)(q.x >> 8) + 1.0f) * 5.9604644775390625e-8f;  // uniform on (0, 1], 24 bits: the survival level
    const float* S = sp.prior_surv;
    int lo = 0, hi = sp.prior_m - 1;  // invariant: S[lo] >= v > S[hi]  (S[0] = 1 >= v)
    if (!(v > S[hi])) return sp.prior_thr + sp.prior_step * (float)hi;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (S[mid] >= v) lo = mid; else hi = mid;
    }
    const float a = S[lo], b = S[hi];
    return sp.prior_thr + sp.prior_step * ((float)lo + (a - v) / (a - b));
}

DEVINL float stats_draw(const StatsParams& sp, float mu, float sd, int64_t row, int64_t sys, uint64_t seed) {
    float t = stats_truncnorm(sp, mu, sd, row, sys, seed);
    if (t >= sp.prior_thr) t = stats_prior_draw(sp, row, sys, seed);
    return t;
}

}  // namespace bnn
