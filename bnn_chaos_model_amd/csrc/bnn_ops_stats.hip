// bnn_ops_stats.hip -- the post-sampling statistics of the evaluation scripts: the numpy-replay kernels (fast_truncnorm,
// prior resampling, min over trios, percentiles), the Philox epilogue on materialised pairs, the streaming quantile sketch.
// One of the translation units of libbnn_chaos_hip.so (bnn_internal.h lists them); entry points declared in include/bnn_chaos_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "bnn_abi_common.h"
#include "bnn_common.hip.h"
#include "bnn_stats.hip.h"

using namespace bnn;

// Per-system percentiles over the draws: one workgroup bitonic-sorts one (system, channel) column of R values in LDS.
struct QuantParams { double q[16]; int nq; };
__global__ __launch_bounds__(256) void bnn_quantiles_kernel(const float* __restrict__ samples, int64_t R, int64_t B, int npad, QuantParams qp,
                                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sv[];
    const int64_t b = blockIdx.x >> 1;
    const int ch = blockIdx.x & 1;
    for (int i = threadIdx.x; i < npad; i += 256) sv[i] = i < R ? samples[((int64_t)i * B + b) * 2 + ch] : __builtin_inff();
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npad; i += 256) {
                int l = i ^ j;
                if (l > i) {
                    float a = sv[i], c = sv[l];
                    bool up = (i & k) == 0;
                    if ((a > c) == up) { sv[i] = c; sv[l] = a; }
                }
            }
            __syncthreads();
        }
    if ((int)threadIdx.x < qp.nq) {
        // numpy 'linear': virtual index q/100*(R-1); lerp(a, b, t) = a + (b-a)*t, evaluated from b's side for t >= 0.5
        double vi = qp.q[threadIdx.x] / 100.0 * (double)(R - 1);
        int64_t lo = (int64_t)floor(vi);
        if (lo > R - 1) lo = R - 1;
        int64_t hi = lo + 1 < R ? lo + 1 : R - 1;
        double t = vi - (double)lo, a = sv[lo], c = sv[hi], d = c - a;
        double v = t >= 0.5 ? c - d * (1.0 - t) : a + d * t;
        out[(b * 2 + ch) * qp.nq + threadIdx.x] = (float)v;
    }
}

// fast_truncnorm: one thread per element, candidates in float64 exactly as numpy forms them; the acceptance test is the
// reference's (:352-358): right = inf -> v > left; left = inf -> v < right; else both
__global__ void bnn_truncnorm_kernel(const float* __restrict__ musd, int64_t n, const double* __restrict__ normals, int nsamp, double left,
                                     double right, uint64_t seed, int64_t id0, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f32x2 ms = *reinterpret_cast<const f32x2*>(musd + 2 * i);
    const double loc = ms.x, scale = ms.y;
    double first = 0.0, pick = 0.0;
    bool found = false;
    for (int s0 = 0; s0 < nsamp && !found; s0 += 4) {
        f32x4 z4 = {0, 0, 0, 0};
        if (!normals) {
            const int64_t el = id0 + i;
            z4 = philox_normal4(TAG_TN | (uint32_t)(s0 >> 2), (uint32_t)el, (uint32_t)((uint64_t)el >> 32), 0u, seed);
        }
        for (int k = 0; k < 4 && s0 + k < nsamp; ++k) {
            const double z = normals ? normals[(int64_t)(s0 + k) * n + i] : (double)z4[k];
            const double v = z * scale + loc;  // rand_out * scale + loc (:347-350); no fma (-ffp-contract=off)
            if (s0 + k == 0) first = v;
            const bool ok = right == INFINITY ? v > left : left == INFINITY ? v < right : (v > left && v < right);
            if (ok) { pick = v; found = true; break; }
        }
    }
    out[i] = (float)(found ? pick : first);  // argmax of an all-False mask is 0 (:360-362)
}

// prior resampling: scipy interp1d(kind='linear') evaluated at u[rank] for every element past the threshold
__global__ void bnn_prior_resample_kernel(float* __restrict__ vals, int64_t n, const int64_t* __restrict__ rank, const double* __restrict__ cum,
                                          const double* __restrict__ edge, int64_t m, const double* __restrict__ u, double thr, uint64_t seed,
                                          int64_t id0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!((double)vals[i] >= thr)) return;
    const int64_t k = rank[i];
    double r;
    if (u) {
        r = u[k];
    } else {  // 53-bit uniform in [0,1) from one Philox block, as numpy builds its doubles: (a >> 5) * 2^26 + (b >> 6)
        const int64_t el = id0 + k;
        uint4 q = philox4x32_10(make_uint4(TAG_U, (uint32_t)el, (uint32_t)((uint64_t)el >> 32), 0u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        r = ((double)(q.x >> 5) * 67108864.0 + (double)(q.y >> 6)) / 9007199254740992.0;
    }
    int64_t lo = 0, hi = m;  // np.searchsorted(cum, r), side='left'
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (cum[mid] < r) lo = mid + 1; else hi = mid;
    }
    int64_t idx = lo < 1 ? 1 : (lo > m - 1 ? m - 1 : lo);
    const double xl = cum[idx - 1], xh = cum[idx], yl = edge[idx - 1], yh = edge[idx];
    const double slope = (yh - yl) / (xh - xl);
    vals[i] = (float)(slope * (r - xl) + yl);
}

__global__ void bnn_group_min_kernel(const float* __restrict__ vals, int64_t n, int group, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = vals[i * group];
    for (int j = 1; j < group; ++j) {
        float w = vals[i * group + j];
        v = (w < v || w != w) ? w : v;  // np.min propagates NaN
        if (v != v) break;
    }
    out[i] = v;
}

// The Philox form of the statistics epilogue on materialised (mu, std) pairs: the same per-evaluation routine as the forward
// kernel's fused tail (bnn_stats.hip.h), one thread per evaluation.
__global__ void bnn_stats_draw_kernel(const float* __restrict__ musd, int64_t R, int64_t B, StatsParams sp, uint64_t seed, int64_t row_id0,
                                      int64_t sys_id0, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * B) return;
    const int64_t r = i / B, b = i % B;
    const f32x2 ms = *reinterpret_cast<const f32x2*>(musd + 2 * i);
    out[i] = stats_draw(sp, ms.x, ms.y, row_id0 + r, sys_id0 + b, seed);
}

// ---- streaming quantile sketch ----------------------------------------------------------------------------------
// Per simulation (= `group` consecutive systems; min over the group first, figures/multiswag_5_planet.py:428) a histogram over
// piecewise-uniform bins plus float64 sum / sum of squares.  hist is bin-major [nbins][n_sims] so that the threads of a wave (one
// simulation each) touch neighbouring words.  Bin 0 collects everything below the first segment (reported as the segment's lower
// edge: only there is the error unbounded; with the scripts' truncation at 4 such a value needs 40 rejected candidates in a row);
// the LAST bin counts NaN draws (a bad seed index poisons its draws): a simulation with any NaN draw gets NaN percentiles, as
// np.percentile would give.
struct SketchSpec {
    int32_t nseg, nbins;
    float lo[4], hi[4], inv_w[4];
    int32_t n[4], base[4];
};

DEVINL int sketch_bin(const SketchSpec& sk, float t) {
    if (t != t) return sk.nbins - 1;
    if (!(t >= sk.lo[0])) return 0;
    int s = 0;
    while (s + 1 < sk.nseg && t >= sk.hi[s]) ++s;
    int k = (int)((t - sk.lo[s]) * sk.inv_w[s]);
    k = k < 0 ? 0 : (k > sk.n[s] - 1 ? sk.n[s] - 1 : k);
    return sk.base[s] + k;
}

__global__ void bnn_sketch_update_kernel(const float* __restrict__ tv, int64_t R, int64_t B, int group, SketchSpec sk, uint32_t* __restrict__ hist,
                                         double* __restrict__ mom) {
    const int64_t n_sims = B / group;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sims) return;
    double s1 = 0.0, s2 = 0.0;
    constexpr int PF = 4;   // rows fetched ahead of their use: the loop is a chain of memory round trips otherwise (R of them per simulation)
    for (int64_t r0 = 0; r0 < R; r0 += PF) {
        float vv[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int64_t r = r0 + u < R ? r0 + u : R - 1;
            const float* p = tv + r * B + i * group;
            float v = p[0];
            for (int j = 1; j < group; ++j) {
                const float w = p[j];
                v = (w < v || w != w) ? w : v;  // np.min propagates NaN
            }
            vv[u] = v;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (r0 + u >= R) break;
            const float v = vv[u];
            // no-return atomic: fire and forget (a plain load-add-store would chain every draw of the slab behind a memory round trip);
            // one thread owns the simulation, so there is no contention
            atomicAdd(&hist[(int64_t)sketch_bin(sk, v) * n_sims + i], 1u);
            s1 += (double)v;
            s2 += (double)v * (double)v;
        }
    }
    mom[2 * i] += s1;
    mom[2 * i + 1] += s2;
}

// numpy 'linear' percentiles from the sketch: order statistic k of a bin holding ranks c .. c+n-1 is placed at
// edge + width * (k - c + 0.5) / n, so every estimate lies in the bin of the exact value (error < one bin width).
struct SketchQ { double q[16]; int nq; };
__global__ void bnn_sketch_quantiles_kernel(const uint32_t* __restrict__ hist, int64_t n_sims, SketchSpec sk, SketchQ qp, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sims) return;
    uint64_t total = 0;
    const int nb = sk.nbins - 1;  // value bins; bin nb counts the NaN draws
    // Both passes fetch the counters EIGHT bins ahead of their use: with one load per iteration and a branch on its value each of the 2 x 945
    // iterations waited out a memory round trip (1.8 ms for 125 000 simulations: latency, not bandwidth).  Same arithmetic, same order.
    constexpr int PF = 8;
    for (int b0 = 0; b0 < nb; b0 += PF) {
        uint32_t v[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) v[j] = hist[(int64_t)(b0 + j < nb ? b0 + j : nb - 1) * n_sims + i];
#pragma unroll
        for (int j = 0; j < PF; ++j) total += (b0 + j < nb) ? v[j] : 0u;
    }
    if (total == 0 || hist[(int64_t)nb * n_sims + i] != 0) {
        for (int k = 0; k < qp.nq; ++k) out[i * qp.nq + k] = __builtin_nanf("");
        return;
    }
    int64_t klo[16];
    double frac[16], vlo[16], vhi[16];
    for (int k = 0; k < qp.nq; ++k) {
        const double vi = qp.q[k] / 100.0 * (double)(total - 1);
        int64_t lo = (int64_t)floor(vi);
        if (lo > (int64_t)total - 1) lo = (int64_t)total - 1;
        klo[k] = lo;
        frac[k] = vi - (double)lo;
        vlo[k] = vhi[k] = 0.0;
    }
    uint64_t c = 0;
    int seg = 0, kin = 0;  // position of bin b inside its segment
    for (int b0 = 0; b0 < nb; b0 += PF) {
        uint32_t v[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) v[j] = hist[(int64_t)(b0 + j < nb ? b0 + j : nb - 1) * n_sims + i];
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int b = b0 + j;
            if (b >= nb) break;
            const uint32_t n = v[j];
            if (!n) continue;   // (an empty bin holds no order statistic; `seg` catches up at the next occupied one)
            double edge, width;
            if (b == 0) { edge = sk.lo[0]; width = 0.0; }
            else {
                while (b >= sk.base[seg] + sk.n[seg]) ++seg;
                kin = b - sk.base[seg];
                width = ((double)sk.hi[seg] - (double)sk.lo[seg]) / (double)sk.n[seg];
                edge = (double)sk.lo[seg] + width * kin;
            }
            for (int k = 0; k < qp.nq; ++k) {
                const int64_t a = klo[k], a1 = (a + 1 < (int64_t)total) ? a + 1 : a;
                if (a >= (int64_t)c && a < (int64_t)(c + n)) vlo[k] = edge + width * ((double)(a - (int64_t)c) + 0.5) / (double)n;
                if (a1 >= (int64_t)c && a1 < (int64_t)(c + n)) vhi[k] = edge + width * ((double)(a1 - (int64_t)c) + 0.5) / (double)n;
            }
            c += n;
        }
    }
    for (int k = 0; k < qp.nq; ++k) out[i * qp.nq + k] = (float)(vlo[k] + (vhi[k] - vlo[k]) * frac[k]);
}

// ---- streaming statistics epilogue ----------------------------------------------------------------------------------
int bnn::stats_params(const bnn_stats* st, StatsParams* sp) {
    if (!st) return fail(BNN_ERR_INVALID, "stats is NULL");
    if (st->tn_nsamp < 1 || st->tn_nsamp > 4096) return fail(BNN_ERR_RANGE, "tn_nsamp must be in [1, 4096]");
    const bool prior = st->prior_thr < INFINITY;
    if (prior && (!st->prior_surv || st->prior_m < 2 || !(st->prior_step > 0.0f))) return fail(BNN_ERR_INVALID, "prior table missing");
    sp->tn_nsamp = st->tn_nsamp; sp->tn_left = st->tn_left; sp->prior_thr = st->prior_thr;
    sp->prior_surv = st->prior_surv; sp->prior_m = st->prior_m; sp->prior_step = st->prior_step;
    return 0;
}

extern "C" {

int bnn_truncnorm_f32(const float* musd, int64_t n, const double* normals, int32_t nsamp, double left, double right, uint64_t philox_seed, int64_t id0,
                      float* out, void* stream) {
    if (n < 0 || nsamp < 1) return fail(BNN_ERR_INVALID, "bad n/nsamp");
    if (n == 0) return 0;
    if (!musd || !out) return fail(BNN_ERR_INVALID, "NULL argument");
    hipLaunchKernelGGL(bnn_truncnorm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, musd, n, normals, (int)nsamp,
                       left, right, philox_seed, id0, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_prior_resample_f32(float* vals, int64_t n, const int64_t* rank, const double* cum, const double* edge, int64_t m, const double* u,
                           double threshold, uint64_t philox_seed, int64_t id0, void* stream) {
    if (n < 0) return fail(BNN_ERR_INVALID, "bad n");
    if (n == 0) return 0;
    if (!vals || !rank || !cum || !edge || m < 2) return fail(BNN_ERR_INVALID, "NULL argument or table shorter than 2");
    hipLaunchKernelGGL(bnn_prior_resample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vals, n, rank, cum, edge,
                       m, u, threshold, philox_seed, id0);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_group_min_f32(const float* vals, int64_t n, int32_t group, float* out, void* stream) {
    if (n < 0 || group < 1) return fail(BNN_ERR_INVALID, "bad n/group");
    if (n == 0) return 0;
    if (!vals || !out) return fail(BNN_ERR_INVALID, "NULL argument");
    hipLaunchKernelGGL(bnn_group_min_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vals, n, (int)group, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_quantiles_f32(const float* samples, int64_t R, int64_t B, const double* host_q, int32_t nq, float* out, void* stream) {
    if (R < 0 || B < 0 || nq < 1 || nq > 16 || !host_q) return fail(BNN_ERR_INVALID, "bad argument");
    if (B == 0) return 0;
    if (R < 1 || R > 16384) return fail(BNN_ERR_RANGE, "quantiles need 1 <= R <= 16384 draws");
    if (!samples || !out) return fail(BNN_ERR_INVALID, "NULL argument");
    QuantParams qp;
    qp.nq = nq;
    for (int i = 0; i < nq; ++i) {
        if (!(host_q[i] >= 0.0 && host_q[i] <= 100.0)) return fail(BNN_ERR_RANGE, "percentiles must be in [0, 100]");
        qp.q[i] = host_q[i];
    }
    int npad = 2;
    while (npad < R) npad <<= 1;
    allow_big_lds<&bnn_quantiles_kernel>();
    if (2 * B > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many systems for one launch");
    hipLaunchKernelGGL(bnn_quantiles_kernel, dim3((unsigned)(2 * B)), dim3(256), (size_t)npad * sizeof(float), (hipStream_t)stream, samples, R,
                       B, npad, qp, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_prior_table_f32(double thr, double top, int32_t m, float* host_surv, double* host_step) {
    if (m < 2 || !(top > thr) || !host_surv || !host_step) return fail(BNN_ERR_INVALID, "bad prior table request");
    // figures/multiswag_5_planet.py:400-404: p(t) ~ a exp(-b t) - c exp(-d t^2); G(t) = integral of p from t to infinity
    const double a = 3.27086190404742, b = 0.424033970670719, c = 10.8793430454878, d = 0.200351029031774;
    auto G = [&](double t) { return a / b * std::exp(-b * t) - c * 0.5 * std::sqrt(M_PI / d) * std::erfc(std::sqrt(d) * t); };
    const double g0 = G(thr), step = (top - thr) / (double)(m - 1);
    for (int i = 0; i < m; ++i) host_surv[i] = (float)(G(thr + step * i) / g0);
    host_surv[0] = 1.0f;
    *host_step = step;
    return 0;
}

int bnn_stats_draw_f32(const float* musd, int64_t R, int64_t B, const bnn_stats* st, uint64_t philox_seed, int64_t row_id0,
                       int64_t system_id0, float* out, void* stream) {
    if (R < 0 || B < 0) return fail(BNN_ERR_INVALID, "bad R/B");
    StatsParams sp;
    int rc = stats_params(st, &sp);
    if (rc) return rc;
    if (R == 0 || B == 0) return 0;
    if (!musd || !out) return fail(BNN_ERR_INVALID, "NULL argument");
    const int64_t n = R * B;
    if ((n + 255) / 256 > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many evaluations for one launch");
    hipLaunchKernelGGL(bnn_stats_draw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, musd, R, B, sp, philox_seed,
                       row_id0, system_id0, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int sketch_spec(const bnn_sketch* sk, SketchSpec* out) {
    if (!sk || sk->nseg < 1 || sk->nseg > 4) return fail(BNN_ERR_INVALID, "sketch needs 1..4 segments");
    SketchSpec s{};
    s.nseg = sk->nseg;
    int base = 1;  // bin 0 = below the first segment
    for (int i = 0; i < sk->nseg; ++i) {
        if (sk->n[i] < 1 || !(sk->hi[i] > sk->lo[i]) || (i && sk->lo[i] != sk->hi[i - 1])) return fail(BNN_ERR_INVALID, "sketch segments must be ascending and contiguous");
        s.lo[i] = sk->lo[i]; s.hi[i] = sk->hi[i]; s.n[i] = sk->n[i]; s.base[i] = base;
        s.inv_w[i] = (float)((double)sk->n[i] / ((double)sk->hi[i] - (double)sk->lo[i]));
        base += sk->n[i];
    }
    s.nbins = base + 1;  // + the NaN counter
    *out = s;
    return 0;
}

int bnn_sketch_bins(const bnn_sketch* sk) {
    SketchSpec s;
    int rc = sketch_spec(sk, &s);
    return rc ? rc : s.nbins;
}

int bnn_sketch_update_u32(const float* t, int64_t R, int64_t B, int32_t group, const bnn_sketch* sk, uint32_t* hist, double* mom, void* stream) {
    SketchSpec s;
    int rc = sketch_spec(sk, &s);
    if (rc) return rc;
    if (R < 0 || B < 0 || group < 1 || (B % group)) return fail(BNN_ERR_INVALID, "B must be a multiple of group");
    if (R == 0 || B == 0) return 0;
    if (!t || !hist || !mom) return fail(BNN_ERR_INVALID, "NULL argument");
    const int64_t n_sims = B / group;
    hipLaunchKernelGGL(bnn_sketch_update_kernel, dim3((unsigned)((n_sims + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, R, B, (int)group, s, hist, mom);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_sketch_quantiles_f32(const uint32_t* hist, int64_t n_sims, const bnn_sketch* sk, const double* host_q, int32_t nq, float* out, void* stream) {
    SketchSpec s;
    int rc = sketch_spec(sk, &s);
    if (rc) return rc;
    if (n_sims < 0 || nq < 1 || nq > 16 || !host_q) return fail(BNN_ERR_INVALID, "bad argument");
    if (n_sims == 0) return 0;
    if (!hist || !out) return fail(BNN_ERR_INVALID, "NULL argument");
    SketchQ qp;
    qp.nq = nq;
    for (int i = 0; i < nq; ++i) {
        if (!(host_q[i] >= 0.0 && host_q[i] <= 100.0)) return fail(BNN_ERR_RANGE, "percentiles must be in [0, 100]");
        qp.q[i] = host_q[i];
    }
    hipLaunchKernelGGL(bnn_sketch_quantiles_kernel, dim3((unsigned)((n_sims + 127) / 128)), dim3(128), 0, (hipStream_t)stream, hist, n_sims, s, qp, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
