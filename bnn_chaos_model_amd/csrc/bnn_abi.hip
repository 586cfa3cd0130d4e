// bnn_abi.hip -- the C ABI of include/bnn_chaos_hip.h plus the small kernels of libbnn_chaos_hip.so (gfx950, MI355X):
// SWAG draw, predictive moments, regress_nn on an explicit summary, the statistics epilogue (numpy-replay forms, the Philox
// form, the streaming quantile sketch), feature packing and the Philox fills.  The forward kernel lives in bnn_forward.hip.h
// and is instantiated by the bnn_fwd_*.hip translation units (bnn_internal.h lists them).
//
// Reference path (MilesCranmer/bnn_chaos_model): SWAGModel.sample_weights + forward_swag_fast / VarModel.forward in
// spock_reg_model.py:415-450, 486-528, 815-908, driven by figures/spock/regression.py:74-92 and
// figures/multiswag_5_planet.py:295-298.  DESIGN.md section 4 has the long form.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (explicit fmaf/MFMA are the only fusions).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bnn_chaos_hip.h"
#include "bnn_common.hip.h"
#include "bnn_stats.hip.h"
#include "bnn_tables.h"

using namespace bnn;

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
// Predictive moments: 64 systems x 16 draw-lanes per workgroup; lane (b, rr) sums draws rr, rr+16, ... in order, the 16
// partials are then added in a fixed tree, so the result is deterministic (but not the strictly sequential sum).
__global__ __launch_bounds__(1024) void bnn_moments_kernel(const float* __restrict__ samples, int64_t R, int64_t B, double* __restrict__ mom,
                                                          int accumulate) {
    __shared__ double part[16][64][4];
    const int l = threadIdx.x, rr = threadIdx.y;
    const int64_t b = (int64_t)blockIdx.x * 64 + l;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (b < B)
        for (int64_t r = rr; r < R; r += 16) {
            f32x2 v = *reinterpret_cast<const f32x2*>(samples + (r * B + b) * 2);
            double mu = v.x, sd = v.y;
            s0 += mu; s1 += mu * mu; s2 += sd; s3 += sd * sd;
        }
    part[rr][l][0] = s0; part[rr][l][1] = s1; part[rr][l][2] = s2; part[rr][l][3] = s3;
    __syncthreads();
    for (int h = 8; h > 0; h >>= 1) {
        if (rr < h)
            for (int k = 0; k < 4; ++k) part[rr][l][k] += part[rr + h][l][k];
        __syncthreads();
    }
    if (rr == 0 && b < B)
        for (int k = 0; k < 4; ++k) mom[b * 4 + k] = (accumulate ? mom[b * 4 + k] : 0.0) + part[0][l][k];
}

// regress_nn + soft_clamp on an explicit summary (predict_instability, spock_reg_model.py:437-442): one thread per system, the
// draw's regress_nn parameters in LDS, each neuron a bias-initialised fmaf chain in the fused kernel's accumulation order, so
// the result is bit-identical to the tail of bnn_forward_f32 on the same summary.
struct RegressParams {
    const float* summary;  // [J,B,SM]  (SM = 40, or 42 with fix_megno)
    const float* W;        // [J,d]
    float* out;            // [J,B,2]
    float* pre;            // [J,B,2] or null
    int64_t B;
    float std_lo, std_span;
    int8_t ord[3][H + 4];
};

constexpr int REG_LD = H + 5;  // odd: conflict-free per-thread rows; holds the 42-wide summary of fix_megno

template <bool MEGNO>
__global__ __launch_bounds__(128) void bnn_regress_kernel(RegressParams p) {
    using Y = Lay<MEGNO>;
    constexpr int NW = Y::D - Y::W4, SM = Y::SM;  // 3362 (3446) floats
    __shared__ float w[NW];
    __shared__ float a[128 * REG_LD];
    __shared__ float h[128 * REG_LD];
    const int tid = threadIdx.x, j = blockIdx.y;
    const float* wj = p.W + (int64_t)j * Y::D + Y::W4;
    for (int i = tid; i < NW; i += 128) w[i] = wj[i];
    const int64_t b = (int64_t)blockIdx.x * 128 + tid;
    const bool live = b < p.B;
    const int64_t o = (int64_t)j * p.B + b;
    for (int k = 0; k < SM; ++k) a[tid * REG_LD + k] = live ? p.summary[o * SM + k] : 0.0f;
    __syncthreads();
    float* av = a + tid * REG_LD;
    float* hv = h + tid * REG_LD;
    for (int n = 0; n < H; ++n) {
        float acc = w[Y::B4 - Y::W4 + n];
        for (int i = 0; i < SM; ++i) { int k = p.ord[0][i]; acc = fmaf(w[n * SM + k], av[k], acc); }
        hv[n] = relu_ieee(acc);
    }
    for (int n = 0; n < H; ++n) {
        float acc = w[Y::B5 - Y::W4 + n];
        for (int i = 0; i < H; ++i) { int k = p.ord[1][i]; acc = fmaf(w[Y::W5 - Y::W4 + n * H + k], hv[k], acc); }
        av[n] = relu_ieee(acc);
    }
    float r[2];
    for (int n = 0; n < 2; ++n) {
        float acc = w[Y::B6 - Y::W4 + n];
        for (int i = 0; i < H; ++i) { int k = p.ord[2][i]; acc = fmaf(w[Y::W6 - Y::W4 + n * H + k], av[k], acc); }
        r[n] = acc;
    }
    if (!live) return;
    float mu = (0.5f * (tanhf(r[0]) + 1.0f)) * 8.0f + 4.0f;
    float sd = (0.5f * (tanhf(r[1]) + 1.0f)) * p.std_span + p.std_lo;
    *reinterpret_cast<f32x2*>(p.out + o * 2) = (f32x2){mu, sd};
    if (p.pre) *reinterpret_cast<f32x2*>(p.pre + o * 2) = (f32x2){r[0], r[1]};
}

// The same for the hparam-built network (generic engine): regress_nn = mlp(SM, 2, hidden, depth_out) with run-time shapes, natural
// accumulation order (bias, then inputs ascending) = the tail of bnn_forward_generic_kernel, bit for bit.  One thread per system,
// activations in registers-per-thread LDS rows, weights straight from the flat vector (L1/L2: every thread of a block reads the
// same address).
struct GenRegressParams {
    const float* summary;  // [J,B,SM]
    const float* W;        // [J,d]
    float* out;
    float* pre;
    int64_t B;
    float std_lo, std_span;
    int32_t d, SM, n_reg, ld;
    GenLayer layer[GEN_MAX_LAYERS / 2 + 1];
};
__global__ __launch_bounds__(64) void bnn_regress_generic_kernel(GenRegressParams p) {
    extern __shared__ float rs[];   // [2][64][ld]
    const int tid = threadIdx.x, j = blockIdx.y;
    const float* wj = p.W + (int64_t)j * p.d;
    const int64_t b = (int64_t)blockIdx.x * 64 + tid;
    const bool live = b < p.B;
    const int64_t o = (int64_t)j * p.B + b;
    float* cur = rs + tid * p.ld;
    float* nxt = rs + (64 + tid) * p.ld;
    for (int k = 0; k < p.SM; ++k) cur[k] = live ? p.summary[o * p.SM + k] : 0.0f;
    for (int l = 0; l < p.n_reg; ++l) {
        const GenLayer ly = p.layer[l];
        for (int n = 0; n < ly.N; ++n) {
            float acc = wj[ly.off_b + n];
            const float* wr = wj + ly.off_w + (int64_t)n * ly.K;
            for (int k = 0; k < ly.K; ++k) acc = fmaf(wr[k], cur[k], acc);
            nxt[n] = ly.relu ? relu_ieee(acc) : acc;
        }
        float* t = cur; cur = nxt; nxt = t;
    }
    if (!live) return;
    const float r0 = cur[0], r1 = cur[1];
    float mu = (0.5f * (tanhf(r0) + 1.0f)) * 8.0f + 4.0f;
    float sd = (0.5f * (tanhf(r1) + 1.0f)) * p.std_span + p.std_lo;
    *reinterpret_cast<f32x2*>(p.out + o * 2) = (f32x2){mu, sd};
    if (p.pre) *reinterpret_cast<f32x2*>(p.pre + o * 2) = (f32x2){r0, r1};
}

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

// data_setup_kernel + StandardScaler.transform + .float() (figures/spock/regression.py:183-213, :144-145).
// HBM-bound by design: 208 B read and 164 B (+ 328 B with X64) written per row.  A workgroup takes PACK_ROWS rows:
//   A. the rows' 26 raw doubles come in with fully coalesced 8-byte loads (one contiguous run per workgroup) into LDS;
//   B. the packed float64 row is built in LDS by DENSE task lists: (row, angle) pairs -- one float64 sincos each, every lane of a wave
//      busy (a thread per raw column left 9 of 32 lanes in the sincos code) -- then (row, plain column) pairs;
//   C. the 41-column rows go out as one contiguous run per workgroup: standardise in float64 (the scaler's own (v - mean) / scale),
//      round to float, coalesced 4-byte stores (8-byte stores for X64).
constexpr int PACK_ROWS = 64;
__constant__ int8_t PACK_ANGLE[9] = {11, 12, 13, 17, 18, 19, 23, 24, 25};   // raw columns expanded to (cos, sin) (regression.py:197-206)
// output column of raw column j (j < 29: the 26 series + 3 masses): every angle before it adds one column; flags follow at 38..40
__host__ __device__ inline int pack_out_col(int j) {
    return j + (j > 11 ? (j < 14 ? j - 11 : 3) : 0) + (j > 17 ? (j < 20 ? j - 17 : 3) : 0) + (j > 23 ? (j < 26 ? j - 23 : 3) : 0);
}
__global__ __launch_bounds__(256) void bnn_feature_pack_kernel(const double* __restrict__ ts, const double* __restrict__ mass, int64_t N, int T,
                                                               const double* __restrict__ mean, const double* __restrict__ scale,
                                                               double* __restrict__ X64, float* __restrict__ x32) {
    __shared__ double raw[PACK_ROWS * 26];
    __shared__ double pk[PACK_ROWS * F];
    __shared__ double ms[2 * F];                                            // the scaler's mean | scale
    const int tid = threadIdx.x;
    const int64_t rows = N * T, row0 = (int64_t)blockIdx.x * PACK_ROWS;
    const int nr = (int)(rows - row0 < PACK_ROWS ? rows - row0 : PACK_ROWS);
    const int64_t n0 = row0 / T;                                            // system of the block's first row (one 64-bit division per block)
    const int t0 = (int)(row0 - n0 * T);
    if (x32 && tid < 2 * F) ms[tid] = tid < F ? mean[tid] : scale[tid - F];
    const double* src = ts + row0 * 26;
    for (int i = tid; i < nr * 26; i += 256) {
        const double v = src[i];
        raw[i] = isfinite(v) ? v : 0.0;                                    // nan_to_num(posinf=0, neginf=0) (:195); the flags below read src again
    }
    __syncthreads();
    for (int t = tid; t < nr * 9; t += 256) {                               // B1: angles
        const int r = t / 9, a = t - 9 * r, j = PACK_ANGLE[a];
        double sn, cs;
        sincos(raw[r * 26 + j], &sn, &cs);
        const int o = pack_out_col(j);
        pk[r * F + o] = cs;
        pk[r * F + o + 1] = sn;
    }
    for (int t = tid; t < nr * 23; t += 256) {                              // B2: 17 plain series columns, 3 masses, 3 flags
        const int r = t / 23, c = t - 23 * r;
        double v;
        int o;
        if (c < 17) {
            const int j = c < 11 ? c : (c < 14 ? c + 3 : c + 6);            // raw columns 0..10, 14..16, 20..22
            v = raw[r * 26 + j];
            o = pack_out_col(j);
        } else if (c < 20) {
            const double m = mass[(n0 + (t0 + r) / T) * 3 + (c - 17)];
            v = isfinite(m) ? m : 0.0;
            o = pack_out_col(26 + (c - 17));
        } else {                                                            // isnotfinite flags of raw columns 3, 6, 7 (:191-193)
            const int j = c == 20 ? 3 : c == 21 ? 6 : 7;
            v = (double)!isfinite(src[r * 26 + j]);
            o = 38 + (c - 20);
        }
        pk[r * F + o] = v;
    }
    __syncthreads();
    for (int i = tid; i < nr * F; i += 256) {                               // C: one contiguous run per workgroup
        const int col = i % F;
        const double v = pk[i];
        if (X64) X64[row0 * F + i] = v;
        if (x32) x32[row0 * F + i] = (float)((v - ms[col]) / ms[F + col]);
    }
}

// Already packed X [N,T,41] float64: standardise only, one thread per element.
__global__ void bnn_standardise_kernel(const double* __restrict__ Xin, int64_t n, const double* __restrict__ mean, const double* __restrict__ scale,
                                       double* __restrict__ X64, float* __restrict__ x32) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % F);
    const double v = Xin[i];
    if (X64) X64[i] = v;
    if (x32) x32[i] = (float)((v - mean[col]) / scale[col]);
}

__global__ void bnn_philox_fill_kernel(int kind, uint64_t seed, int64_t id0, int64_t n_rows, int64_t B, int64_t sys0, int width, int aux,
                                       float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (kind == 0 || kind == 1) {
        int64_t total = n_rows * width;
        if (i >= total) return;
        int64_t row = i / width;
        int el = (int)(i % width);
        out[i] = philox_z(kind == 0 ? TAG_Z1 : TAG_Z2, id0 + row, el, seed);
    } else if (kind == 2 || kind == 4) {   // eps [n_rows,B,2,L] with L = width (0: 20); eps_sum [n_rows,B,SM] with SM = width (0: 40)
        const int SM = kind == 4 ? (width > 0 ? width : S2) : 2 * (width > 0 ? width : L);
        int64_t total = n_rows * B * SM;
        if (i >= total) return;
        int el = (int)(i % SM);
        int64_t sys = (i / SM) % B, row = i / ((int64_t)SM * B);
        out[i] = philox_sys4(kind == 2 ? TAG_EPS : TAG_SUM, id0 + row, sys0 + sys, el >> 2, seed)[el & 3];
    } else if (kind == 5) {  // candidates of the truncated-normal draw [n_rows, B, nsamp = width] (bnn_stats.hip.h)
        int64_t total = n_rows * B * width;
        if (i >= total) return;
        int k = (int)(i % width);
        int64_t sys = (i / width) % B, row = i / ((int64_t)width * B);
        out[i] = philox_sys4(TAG_TNS, id0 + row, sys0 + sys, k >> 2, seed)[k & 3];
    } else if (kind == 6) {  // survival level of the prior draw [n_rows, B], uniform on (0, 1]
        int64_t total = n_rows * B;
        if (i >= total) return;
        int64_t sys = i % B, row = i / B;
        const uint4 q = philox4x32_10(philox_sys_ctr(TAG_US, id0 + row, sys0 + sys, 0), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        out[i] = ((float)(q.x >> 8) + 1.0f) * 5.9604644775390625e-8f;
    } else {  // kind 3: eps_in [n_rows, B, T = width, NF = aux (0: 41)]: block t * ceil(NF/6) + col/6, normal col%6 (bnn_common.hip.h)
        const int T = width, NF = aux > 0 ? aux : F, nblk = (NF + NIN_PER_BLOCK - 1) / NIN_PER_BLOCK;
        int64_t per = (int64_t)T * NF, total = n_rows * B * per;
        if (i >= total) return;
        int col = (int)(i % NF), t = (int)((i / NF) % T);
        int64_t sys = (i / per) % B, row = i / (per * B);
        float n6[6];
        philox_in6(id0 + row, sys0 + sys, t * nblk + col / NIN_PER_BLOCK, seed, n6);
        const int j = col % NIN_PER_BLOCK;
        out[i] = j == 0 ? n6[0] : j == 1 ? n6[1] : j == 2 ? n6[2] : j == 3 ? n6[3] : j == 4 ? n6[4] : n6[5];
    }
}

__global__ void bnn_philox_raw_kernel(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int64_t n,
                                      uint32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 r = philox4x32_10(make_uint4(c0 + (uint32_t)i, c1, c2, c3), make_uint2(k0, k1));
    out[i * 4] = r.x; out[i * 4 + 1] = r.y; out[i * 4 + 2] = r.z; out[i * 4 + 3] = r.w;
}

// SWAGModel.sample_weights for J draws: grid.x = (draw, 256-row slice of the parameter vector), so J is bounded only by 2^31 / 30.
// D = length of the flat parameter vector (7583; 7665 with fix_megno).
__global__ __launch_bounds__(256) void bnn_swag_draw_kernel(const float* __restrict__ w_avg, const float* __restrict__ w2_avg,
                                                            const float* __restrict__ pre_D, int D, int S, int K,
                                                            const int32_t* __restrict__ seed_idx, const float* __restrict__ z1,
                                                            const float* __restrict__ z2, float c1, float c2, float scale,
                                                            uint64_t seed, int64_t draw_id0, float* __restrict__ W_out) {
    __shared__ float slabs[4 * SLAB];
    __shared__ float zsh[MAXK_DRAW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int DRAW_SLICES = (D + 255) / 256;
    const int64_t e = blockIdx.x / DRAW_SLICES;
    const int slice = blockIdx.x % DRAW_SLICES;
    int s = seed_idx[e];
    const bool bad = (s < 0 || s >= S);
    if (bad) s = 0;
    for (int k = threadIdx.x; k < K; k += 256) zsh[k] = z2 ? z2[e * K + k] : philox_z(TAG_Z2, draw_id0 + e, k, seed);
    const int i0 = (slice * 4 + wave) * 64;
    const float* pd = pre_D + (int64_t)s * D * K;
    const int i = i0 + lane;
    const bool live = i < D;
    const float wa = live ? w_avg[(int64_t)s * D + i] : 0.0f;
    float w = 0.0f, dot = 0.0f;
    if (live) {
        const float z1v = z1 ? z1[e * (int64_t)D + i] : philox_z(TAG_Z1, draw_id0 + e, i, seed);
        w = draw_head(wa, w2_avg[(int64_t)s * D + i], z1v, c1);
    }
    for (int kc = 0; kc < K; kc += MAXK) {   // the deviation columns, 32 at a time through the slab; the dot product runs on in k order
        const int Kc = K - kc < MAXK ? K - kc : MAXK;
        if (kc) __syncthreads();
        if (i0 < D) draw_stage(pd, i0, D, K, kc, Kc, lane, slabs + wave * SLAB);
        __syncthreads();
        if (live) dot = draw_dot(slabs + wave * SLAB + lane * Kc, wa, zsh + kc, Kc, dot);
    }
    if (live) W_out[e * (int64_t)D + i] = bad ? __builtin_nanf("") : draw_finish(w, dot, c2, scale);
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
    for (int64_t r = 0; r < R; ++r) {
        const float* p = tv + r * B + i * group;
        float v = p[0];
        for (int j = 1; j < group; ++j) {
            const float w = p[j];
            v = (w < v || w != w) ? w : v;  // np.min propagates NaN
        }
        // no-return atomic: fire and forget (a plain load-add-store would chain every draw of the slab behind a memory round trip);
        // one thread owns the simulation, so there is no contention
        atomicAdd(&hist[(int64_t)sketch_bin(sk, v) * n_sims + i], 1u);
        s1 += (double)v;
        s2 += (double)v * (double)v;
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
    for (int b = 0; b < nb; ++b) total += hist[(int64_t)b * n_sims + i];
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
    for (int b = 0; b < nb; ++b) {
        const uint32_t n = hist[(int64_t)b * n_sims + i];
        double edge, width;
        if (b == 0) { edge = sk.lo[0]; width = 0.0; }
        else {
            while (b >= sk.base[seg] + sk.n[seg]) ++seg;
            kin = b - sk.base[seg];
            width = ((double)sk.hi[seg] - (double)sk.lo[seg]) / (double)sk.n[seg];
            edge = (double)sk.lo[seg] + width * kin;
        }
        if (n) {
            for (int k = 0; k < qp.nq; ++k) {
                const int64_t a = klo[k], a1 = (a + 1 < (int64_t)total) ? a + 1 : a;
                if (a >= (int64_t)c && a < (int64_t)(c + n)) vlo[k] = edge + width * ((double)(a - (int64_t)c) + 0.5) / (double)n;
                if (a1 >= (int64_t)c && a1 < (int64_t)(c + n)) vhi[k] = edge + width * ((double)(a1 - (int64_t)c) + 0.5) / (double)n;
            }
        }
        c += n;
    }
    for (int k = 0; k < qp.nq; ++k) out[i * qp.nq + k] = (float)(vlo[k] + (vhi[k] - vlo[k]) * frac[k]);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) return fail(BNN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct bnn_plan {
    bnn_arch arch;
    bool megno = false;  // arch.fix_megno
    bool v50net = false; // the pretrained ensemble's network: 41 -> 40 -> 40 -> 20 / 40 (42) -> 40 -> 40 -> 2 (register-resident kernels)
    int d = D;           // length of the flat parameter vector
    Tables tab[2];  // [0] = arch mask, [1] = all 41 columns (noisy forward)      (v50net only)
    int16_t* d_f2 = nullptr;   // regress_nn fragment gather table
    int16_t* d_f4 = nullptr;   // feature_nn weight-register table (4x4x1 path) for the plan's mask
    int16_t* d_f4n = nullptr;  // ... with every column live (noisy forward)
    float* d_rcp = nullptr;    // [RCP_N] 1/(i+1)
    GenArch gen;               // generic engine: every plan has one (the v50 network falls back to it for T % 4 != 0 or T < 8)
    GenArch* d_gen = nullptr;
    // specialised forms of the generic engine, compiled at run time for this network (bnn_spec_source / bnn_plan_attach_spec): [noisy]
    GenArch spec_gen[2];
    hipModule_t spec_mod[2] = {nullptr, nullptr};
    hipFunction_t spec_fn[2] = {nullptr, nullptr};
    bool emb[2] = {false, false};   // the pretrained network's specialised forms compiled into the library (bnn_fwd_v50spec.hip) apply: [noisy]
    GenArch emb_gen[2];
    int device = 0;
};

static bool is_v50net(const bnn_arch* a) {
    return a->n_features == F && a->hidden == H && a->latent == L && a->depth_in == 1 && a->depth_out == 1;
}

static int check_arch(const bnn_arch* a, GenArch* gen_out = nullptr) {
    if (!a) return fail(BNN_ERR_INVALID, "arch is NULL");
    if (a->fix_megno != 0 && a->fix_megno != 1) return fail(BNN_ERR_INVALID, "fix_megno must be 0 or 1");
    GenArch g;
    const char* why = "";
    if (gen_build(a->n_features, a->hidden, a->latent, a->depth_in, a->depth_out, a->fix_megno != 0, &g, &why))
        return fail(BNN_ERR_UNSUPPORTED, why);
    if (a->n_features < 64 && (a->zero_mask >> a->n_features)) return fail(BNN_ERR_INVALID, "zero_mask has bits beyond the last column");
    if (gen_out) *gen_out = g;
    return 0;
}

extern "C" {

int bnn_abi_version(void) { return BNN_ABI_VERSION; }
const char* bnn_last_error(void) { return g_err.c_str(); }

int bnn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(BNN_ERR_NO_DEVICE, hipGetErrorString(e));
    return n;
}

#ifndef BNN_BUILD_FLAGS
#define BNN_BUILD_FLAGS ""
#endif
const char* bnn_build_flags(void) { return BNN_BUILD_FLAGS; }

int bnn_param_count(const bnn_arch* arch) {
    GenArch g;
    int rc = check_arch(arch, &g);
    return rc ? rc : g.d;
}

static bool upload(const void* host, size_t bytes, void** dev) {
    return hipMalloc(dev, bytes) == hipSuccess && hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice) == hipSuccess;
}

int bnn_plan_create(const bnn_arch* arch, bnn_plan** out) {
    GenArch g;
    int rc = check_arch(arch, &g);
    if (rc) return rc;
    if (!out) return fail(BNN_ERR_INVALID, "out is NULL");
    bnn_plan* pl = new bnn_plan();
    pl->arch = *arch;
    pl->megno = arch->fix_megno != 0;
    pl->v50net = is_v50net(arch);
    pl->gen = g;
    pl->d = g.d;
    if (pl->v50net) {
        pl->tab[0] = build_tables(arch->zero_mask, false, pl->megno);
        pl->tab[1] = build_tables(arch->zero_mask, true, pl->megno);
        if (pl->d != layout_of(pl->megno).D) { delete pl; return fail(BNN_ERR_INVALID, "internal: the two engines disagree on the parameter count"); }
    }
    if (pl->v50net && !pl->megno) {   // the forms of bnn_fwd_v50spec.hip: noisy under any mask, quiet under the pretrained one
        const char* why = "";
        pl->emb[1] = gen_build_spec(F, H, L, 1, 1, false, 1, 0, 1, &pl->emb_gen[1], &why) == 0;
        pl->emb[0] = arch->zero_mask == V50_ZERO_MASK && gen_build_spec(F, H, L, 1, 1, false, 1, V50_ZERO_MASK, 1, &pl->emb_gen[0], &why) == 0;
    }
    if (hipGetDevice(&pl->device) != hipSuccess) {
        delete pl;
        return fail(BNN_ERR_NO_DEVICE, "no HIP device");
    }
    std::vector<float> rcp(RCP_N);
    for (int i = 0; i < RCP_N; ++i) rcp[i] = 1.0f / (float)(i + 1);
    bool ok = upload(rcp.data(), RCP_N * sizeof(float), (void**)&pl->d_rcp) && upload(&pl->gen, sizeof(GenArch), (void**)&pl->d_gen);
    if (ok && pl->v50net)
        ok = upload(pl->tab[0].f2.data(), pl->tab[0].f2.size() * sizeof(int16_t), (void**)&pl->d_f2) &&
             upload(pl->tab[0].f4.data(), pl->tab[0].f4.size() * sizeof(int16_t), (void**)&pl->d_f4) &&
             upload(pl->tab[1].f4.data(), pl->tab[1].f4.size() * sizeof(int16_t), (void**)&pl->d_f4n);
    if (!ok) {
        bnn_plan_destroy(pl);
        return fail(BNN_ERR_HIP, "plan table upload failed");
    }
    *out = pl;
    return 0;
}

int bnn_plan_destroy(bnn_plan* pl) {
    if (!pl) return 0;
    if (pl->d_f2) (void)hipFree(pl->d_f2);
    if (pl->d_f4) (void)hipFree(pl->d_f4);
    if (pl->d_f4n) (void)hipFree(pl->d_f4n);
    if (pl->d_rcp) (void)hipFree(pl->d_rcp);
    if (pl->d_gen) (void)hipFree(pl->d_gen);
    for (int i = 0; i < 2; ++i)
        if (pl->spec_mod[i]) (void)hipModuleUnload(pl->spec_mod[i]);
    delete pl;
    return 0;
}

// the descriptor of one specialised form: the quiet kq-major forms drop the plan's masked input columns from layer 0
static int spec_arch(const bnn_arch* a, int32_t w8, int32_t noisy, int32_t flags, GenArch* g) {
    if (flags & ~(BNN_SPEC_POOL_REGS | BNN_SPEC_BLOCK_MAJOR | BNN_SPEC_RESIDENT)) return fail(BNN_ERR_INVALID, "unknown specialisation flag");
    if ((flags & BNN_SPEC_RESIDENT) && (flags & BNN_SPEC_BLOCK_MAJOR)) return fail(BNN_ERR_INVALID, "BNN_SPEC_RESIDENT belongs to the input-quad-major forms");
    if (noisy != 0 && noisy != 1) return fail(BNN_ERR_INVALID, "noisy must be 0 or 1");
    const char* why = "";
    const uint64_t drop = (noisy || (flags & BNN_SPEC_BLOCK_MAJOR)) ? 0 : a->zero_mask;
    if (gen_build_spec(a->n_features, a->hidden, a->latent, a->depth_in, a->depth_out, a->fix_megno != 0, w8, drop, (flags & BNN_SPEC_POOL_REGS) ? 1 : 0, g, &why))
        return fail(BNN_ERR_UNSUPPORTED, why);
    return 0;
}

int bnn_spec_embedded_source(char* buf, size_t cap) {
    const int n = gen_spec_embedded_source(buf, cap);
    return n < 0 ? fail(BNN_ERR_UNSUPPORTED, "internal: the pretrained network's specialised descriptor") : n;
}

int bnn_spec_source(const bnn_arch* arch, int32_t w8, int32_t noisy, int32_t flags, char* buf, size_t cap) {
    int rc = check_arch(arch);
    if (rc) return rc;
    GenArch g;
    rc = spec_arch(arch, w8, noisy, flags, &g);
    if (rc) return rc;
    if (flags & BNN_SPEC_RESIDENT) {   // a few dozen registers at most: beyond that nothing is left for the activations
        int nw = 0;
        for (int l = 0; l < g.n_feat; ++l) nw += g.layer[l].nkq * g.layer[l].nblk;
        if (nw > 112) return fail(BNN_ERR_UNSUPPORTED, "feature_nn's weight registers do not fit next to the activations (more than 112)");
    }
    return gen_spec_source(g, noisy, (flags & BNN_SPEC_POOL_REGS) ? 1 : 0, (flags & BNN_SPEC_BLOCK_MAJOR) ? 1 : 0,
                           g.in_live ? arch->zero_mask : 0, buf, cap, (flags & BNN_SPEC_RESIDENT) ? 1 : 0);
}

int bnn_plan_attach_spec(bnn_plan* pl, int32_t noisy, int32_t w8, int32_t flags, const void* image, size_t bytes) {
    if (!pl || !image || !bytes) return fail(BNN_ERR_INVALID, "plan/image is NULL");
    GenArch g;
    int rc = spec_arch(&pl->arch, w8, noisy, flags, &g);
    if (rc) return rc;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != pl->device) return fail(BNN_ERR_INVALID, "the plan lives on another device than the current one");
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    hipError_t e = hipModuleLoadData(&mod, image);
    if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("hipModuleLoadData: ") + hipGetErrorString(e));
    e = hipModuleGetFunction(&fn, mod, "bnn_spec_forward");
    if (e != hipSuccess) {
        (void)hipModuleUnload(mod);
        return fail(BNN_ERR_HIP, std::string("the code object has no kernel bnn_spec_forward: ") + hipGetErrorString(e));
    }
    if (pl->spec_mod[noisy]) (void)hipModuleUnload(pl->spec_mod[noisy]);
    pl->spec_mod[noisy] = mod;
    pl->spec_fn[noisy] = fn;
    pl->spec_gen[noisy] = g;
    return 0;
}

int bnn_plan_spec_attached(const bnn_plan* pl, int32_t noisy) {
    if (!pl || (noisy != 0 && noisy != 1)) return fail(BNN_ERR_INVALID, "plan is NULL / noisy must be 0 or 1");
    return pl->spec_fn[noisy] ? 1 : 0;
}

size_t bnn_gen_params_bytes(void) { return sizeof(GenParams); }

size_t bnn_nonfinite_record_bytes(int64_t B) { return sizeof(int32_t) * (size_t)(4 + (B > 0 ? B : 0)); }

int bnn_nonfinite_scan_f32(const bnn_plan* pl, const float* x, int64_t B, int32_t T, void* record, void* stream) {
    if (!pl) return fail(BNN_ERR_INVALID, "plan is NULL");
    if (B < 0 || T < 1) return fail(BNN_ERR_INVALID, "bad B/T");
    if (!record) return fail(BNN_ERR_INVALID, "record is NULL");
    if (B >= (1LL << 30)) return fail(BNN_ERR_RANGE, "the scan record indexes at most 2^30 systems per call: shard the batch");
    if (B > 0 && !x) return fail(BNN_ERR_INVALID, "x is NULL");
    // a masked column holds NaN after `x - mask` whichever non-finite value it had; fix_megno zeroes the MEGNO column whatever the
    // mask says (:488-491) and summarises the raw one (:480-484): non-finite there is NaN in the summary either way
    const uint64_t mask = pl->arch.zero_mask | (pl->megno ? (1ull << MEGNO_COL) : 0ull);
    hipError_t e = launch_nonfinite_scan(x, B, (int64_t)T * pl->arch.n_features, pl->arch.n_features, mask, static_cast<int32_t*>(record), (hipStream_t)stream);
    if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("non-finite scan: ") + hipGetErrorString(e));
    return 0;
}

// natural order of the generic engine: the live inputs ascending (layer 0 drops the masked columns unless `noisy`)
static int natural_order(const GenArch& g, uint64_t zero_mask, int layer, int noisy, int32_t* host_order, int cap) {
    if (layer < 0 || layer >= g.n_feat + g.n_reg) return fail(BNN_ERR_INVALID, "layer index beyond the network's Linear modules");
    int n = 0;
    for (int k = 0; k < g.layer[layer].K; ++k) {
        if (layer == 0 && !noisy && k < 64 && ((zero_mask >> k) & 1ull)) continue;
        if (host_order && n < cap) host_order[n] = k;
        ++n;
    }
    return n;
}

int bnn_plan_layer_order(const bnn_plan* pl, int layer, int noisy, int32_t* host_order, int cap) {
    if (!pl) return fail(BNN_ERR_INVALID, "plan is NULL");
    if (!pl->v50net) return natural_order(pl->gen, pl->arch.zero_mask, layer, noisy, host_order, cap);
    if (layer < 0 || layer > 5) return fail(BNN_ERR_INVALID, "bad plan/layer");
    const std::vector<int32_t>& o = pl->tab[noisy ? 1 : 0].order[layer];
    if (host_order)
        for (int i = 0; i < (int)o.size() && i < cap; ++i) host_order[i] = o[i];
    return (int)o.size();
}

int bnn_layer_order(const bnn_arch* arch, int layer, int noisy, int32_t* host_order, int cap) {
    GenArch g;
    int rc = check_arch(arch, &g);
    if (rc) return rc;
    if (!is_v50net(arch)) return natural_order(g, arch->zero_mask, layer, noisy, host_order, cap);
    if (layer < 0 || layer > 5) return fail(BNN_ERR_INVALID, "layer must be 0..5");
    Tables t = build_tables(arch->zero_mask, noisy != 0, arch->fix_megno != 0);
    const std::vector<int32_t>& o = t.order[layer];
    if (host_order)
        for (int i = 0; i < (int)o.size() && i < cap; ++i) host_order[i] = o[i];
    return (int)o.size();
}

int bnn_fragment_table(const bnn_arch* arch, int noisy, int which, int16_t* host_table, int cap) {
    int rc = check_arch(arch);
    if (rc) return rc;
    if (!is_v50net(arch)) return fail(BNN_ERR_UNSUPPORTED, "fragment tables belong to the pretrained network's kernels; the generic engine builds its LDS image in the kernel prologue");
    if (which != 1 && which != 2) return fail(BNN_ERR_INVALID, "which must be 1 (feature_nn images) or 2 (regress_nn fragments)");
    Tables t = build_tables(arch->zero_mask, noisy != 0, arch->fix_megno != 0);
    const std::vector<int16_t>& v = which == 1 ? t.f4 : t.f2;
    if (host_table)
        for (int i = 0; i < (int)v.size() && i < cap; ++i) host_table[i] = v[i];
    return (int)v.size();
}

static int pick_spc(const bnn_grid* g, int64_t csz, bool xcd_order) {
    if (g->systems_per_block > 0) return g->systems_per_block;
    // The per-workgroup prologue (flat vector -> weight registers, regress_nn fragments) is amortised over the block: prefer big
    // blocks (512 systems: +1 % over 128 at configs[1]) as long as the grid still fills 256 CUs x 2 several times over.  Under the
    // XCD work order the block is also the L2's working set across draws: 256-system blocks (4.2 MB) reach HBM 79 GB per configs[2]
    // launch against 168 GB for 512 (8.4 MB) -- but HBM is at 3.5 % of its roof either way, the kernel time is the same within
    // 0.3 %, and 256 costs 1.2 % more cycles (twice the prologues; profiles/r03_spb_traffic.txt): 512 stays.
    (void)xcd_order;
    for (int spc : {512, 256, 128}) {
        int64_t nsub = (csz + spc - 1) / spc;
        if (nsub * (int64_t)g->J >= 4096) return spc;
    }
    return 64;
}

static GenMerge gen_merge_consts(int na, int nb) {   // oracle/bnn_oracle.c merge_consts, constant for constant
    GenMerge m{2, 0.0f, 0.0f};
    if (nb == 0) m.mode = 2;
    else if (na == 0) m.mode = 3;
    else if (na == nb) { m.mode = 0; m.w1 = (float)na * 0.5f; }
    else {
        m.mode = 1;
        m.w1 = (float)((double)nb / (double)(na + nb));
        m.w2 = (float)((double)na * (double)nb / (double)(na + nb));
    }
    return m;
}

static int launch_forward(const bnn_plan* pl, const bnn_grid* g, FwdParams& p, bool fused, bool noisy, void* stream, int lowp = 0) {
    if (!pl || !g) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (g->B < 0 || g->J < 0 || g->nchunks < 1) return fail(BNN_ERR_INVALID, "negative size");
    if (g->J % g->nchunks) return fail(BNN_ERR_INVALID, "J must be a multiple of nchunks");
    if (g->T < 2 || (g->T + 3) / 4 > RCP_N) return fail(BNN_ERR_UNSUPPORTED, "T must be in [2, 16384] (torch.std of a single timestep is NaN)");
    if (g->systems_per_block < 0 || (g->systems_per_block % 64)) return fail(BNN_ERR_INVALID, "systems_per_block must be a multiple of 64");
    // the pretrained network's kernels take whole tiles of 4 timesteps and at least two of them; everything else is the generic engine's
    if (g->engine < 0 || g->engine > 2) return fail(BNN_ERR_INVALID, "grid.engine must be 0 (choose), 1 (generic) or 2 (specialised)");
    const bool generic = !pl->v50net || (g->T % 4) != 0 || g->T < 8 || g->engine != 0;
    // the network's own compiled form of the generic engine, when one is attached (engine 2 insists on it)
    const int nz = noisy ? 1 : 0;
    const bool spec_mod = generic && g->engine != 1 && pl->spec_fn[nz] != nullptr;
    const bool spec_emb = generic && g->engine != 1 && !spec_mod && pl->emb[nz];   // (an attached form wins over the embedded one)
    const bool spec = spec_mod || spec_emb;
    if (g->engine == 2 && !spec) return fail(BNN_ERR_UNSUPPORTED, "grid.engine = 2: no specialised form is attached to this plan (bnn_plan_attach_spec)");
    if (generic && lowp) return fail(BNN_ERR_UNSUPPORTED, "the reduced-precision kernels are built for the pretrained network at T % 4 == 0 only");
    if (generic && fused) return fail(BNN_ERR_UNSUPPORTED, "the in-prologue draw (W_workspace = NULL) exists for the pretrained network at T % 4 == 0 only: pass a [J, d] workspace");
    if (g->B == 0 || g->J == 0) return 0;
    if (!p.x || !(p.out || p.sink || p.latents)) return fail(BNN_ERR_INVALID, "x/out is NULL");
    if (p.draw_id0 % g->nchunks) return fail(BNN_ERR_INVALID, "draw_id0 must be a multiple of nchunks");
    const int NF = pl->arch.n_features;
    p.B = g->B; p.T = g->T; p.ntiles = (g->T + 3) / 4; p.J = g->J; p.nch = g->nchunks;
    p.cB = g->chunk_B > 0 ? g->chunk_B : g->B;
    p.coff = g->chunk_B > 0 ? g->chunk_off : 0;
    if (p.coff < 0 || p.coff + g->B > p.cB) return fail(BNN_ERR_INVALID, "chunk_off / chunk_B: the shard [chunk_off, chunk_off + B) must lie inside the batch");
    p.csz = (p.cB + g->nchunks - 1) / g->nchunks;
    p.xcd_order = (g->nchunks > 1 || (double)g->B * g->T * NF * sizeof(float) > 256.0 * 1024 * 1024) ? 1 : 0;
    const int64_t cseg = p.csz < g->B ? p.csz : g->B;   // the longest stretch of one chunk inside this call's rows
    p.spc = pick_spc(g, cseg, p.xcd_order != 0);
    p.row_id0 = p.draw_id0 / g->nchunks;
    p.tab_f2 = pl->d_f2; p.tab_wr = noisy ? pl->d_f4n : pl->d_f4; p.rcp_tab = pl->d_rcp;
    p.zero_mask = pl->arch.zero_mask;
    p.std_lo = pl->arch.lowest_std; p.std_span = (float)(6.0 - (double)pl->arch.lowest_std);
    const int64_t nsub = (cseg + p.spc - 1) / p.spc;
    const int64_t nblk = nsub * g->J;
    if (nblk > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "grid too large; split the draws");
    static_assert(Lay<true>::NF2 * 64 <= FLAT_LDS, "regress_nn fragments overwrite the flat vector in place");
    hipStream_t st = (hipStream_t)stream;
    auto launch = [&]() -> int {
        hipError_t e;
        if (generic) {
            GenParams P{};
            P.f = p;
            P.g = pl->d_gen;
            int cnt[4];
            for (int q = 0; q < 4; ++q) cnt[q] = q < g->T ? (g->T - q + 3) / 4 : 0;   // timesteps t = q, q + 4, ... below T
            P.m01 = gen_merge_consts(cnt[0], cnt[1]);
            P.m23 = gen_merge_consts(cnt[2], cnt[3]);
            P.m0123 = gen_merge_consts(cnt[0] + cnt[1], cnt[2] + cnt[3]);
            P.noisy = noisy ? 1 : 0;
            if (spec_emb) {
                e = launch_fwd_v50spec(noisy, (unsigned)nblk, st, P);
                if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("embedded specialised forward kernel launch: ") + hipGetErrorString(e));
                return 0;
            }
            if (spec_mod) {
                size_t psz = sizeof(P);
                void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &P, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psz, HIP_LAUNCH_PARAM_END};
                e = hipModuleLaunchKernel(pl->spec_fn[noisy ? 1 : 0], (unsigned)nblk, 1, 1, 64u * pl->spec_gen[noisy ? 1 : 0].nwaves, 1, 1, 0, st, nullptr, cfg);
                if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("specialised forward kernel launch: ") + hipGetErrorString(e));
                return 0;
            }
            e = launch_fwd_generic(pl->gen, (unsigned)nblk, st, P);
            if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("generic forward kernel launch: ") + hipGetErrorString(e));
            return 0;
        }
        const bool k31 = pl->tab[0].kin4 == 31;
        if (pl->megno && (lowp || p.sink)) return fail(BNN_ERR_UNSUPPORTED, "fix_megno: the reduced-precision and fused-statistics forms are not built");
        if (pl->megno) e = launch_fwd_megno(k31, fused, noisy, (unsigned)nblk, st, p);
        else if (lowp) e = launch_fwd_lowp(lowp, (unsigned)nblk, st, p);
        else if (p.sink) e = launch_fwd_stats(k31, (unsigned)nblk, st, p);
        else if (noisy) e = launch_fwd_noisy((unsigned)nblk, st, p);
        else if (k31) e = launch_fwd_k31(fused, (unsigned)nblk, st, p);
        else e = launch_fwd_k41(fused, (unsigned)nblk, st, p);
        if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("forward kernel launch: ") + hipGetErrorString(e));
        return 0;
    };
    int rc = launch();
    if (rc || !g->nonfinite) return rc;
    // the systems the scan listed (non-finite inputs) are re-evaluated the reference's way and overwrite what the kernel above wrote for
    // them (bnn_nonfinite.hip); with an empty list the launch leaves at once
    NfxParams q{};
    q.f = p;
    q.g = pl->d_gen;
    q.rec = static_cast<const int32_t*>(g->nonfinite);
    q.noisy = noisy ? 1 : 0;
    q.shortcut = (p.summary || p.latents) ? 0 : 1;
    hipError_t e = launch_nonfinite_fixup(pl->gen, q, st);
    if (e != hipSuccess) return fail(BNN_ERR_HIP, std::string("non-finite fix-up kernel launch: ") + hipGetErrorString(e));
    return 0;
}

static int draw_consts(int K, float scale, float* c1, float* c2, int kmax = MAXK_DRAW) {
    if (K < 2 || K > kmax) return fail(BNN_ERR_RANGE, kmax == MAXK ? "SWAG rank K above 32 needs the draw-once form (W_workspace): the in-prologue draw takes K in [2, 32]"
                                                                  : "SWAG rank K must be in [2, 256]");
    *c1 = (float)((double)scale * (1.0 / std::sqrt(2.0)));  // scale * (1.0/np.sqrt(2.0)), :834
    *c2 = (float)std::sqrt(2.0 * (K - 1));                   // np.sqrt(2*(K-1)), :835
    return 0;
}

int bnn_swag_draw_f32(const bnn_plan* plan, const float* w_avg, const float* w2_avg, const float* pre_D, int32_t S, int32_t K,
                      const int32_t* seed_idx, int32_t J, const float* z1, const float* z2, float scale, uint64_t philox_seed,
                      int64_t draw_id0, float* W_out, void* stream) {
    if (J == 0) return 0;
    if (!plan || !w_avg || !w2_avg || !pre_D || !seed_idx || !W_out) return fail(BNN_ERR_INVALID, "NULL argument");
    if ((z1 == nullptr) != (z2 == nullptr)) return fail(BNN_ERR_INVALID, "z1 and z2 must both be given or both be NULL");
    if (S < 1 || J < 0) return fail(BNN_ERR_INVALID, "bad S/J");
    float c1, c2;
    int rc = draw_consts(K, scale, &c1, &c2);
    if (rc) return rc;
    if (J == 0) return 0;
    const int d = plan->d, slices = (d + 255) / 256;
    if ((int64_t)J * slices > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many draws for one launch");
    dim3 grid((unsigned)((int64_t)J * slices)), block(256);
    hipLaunchKernelGGL(bnn_swag_draw_kernel, grid, block, 0, (hipStream_t)stream, w_avg, w2_avg, pre_D, d, S, K, seed_idx, z1, z2, c1,
                       c2, scale, philox_seed, draw_id0, W_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_forward_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* W, const float* eps, const float* eps_in,
                    const float* eps_sum, uint64_t philox_seed, int64_t draw_id0, int64_t system_id0, float* out, float* pre_clamp,
                    float* summary, void* stream) {
    if (grid && (grid->B == 0 || grid->J == 0)) return 0;
    if (!W) return fail(BNN_ERR_INVALID, "W is NULL");
    if ((eps_in == nullptr) != (eps_sum == nullptr)) return fail(BNN_ERR_INVALID, "eps_in and eps_sum must both be given or both be NULL");
    if (eps_in && !eps) return fail(BNN_ERR_INVALID, "explicit noise is all-or-nothing: eps_in/eps_sum need eps as well");
    const bool noisy = eps_in != nullptr || (grid && grid->noisy);
    if (noisy && !eps_in && eps) return fail(BNN_ERR_INVALID, "explicit noise is all-or-nothing: grid.noisy with eps but no eps_in/eps_sum");
    FwdParams p{};
    p.x = x; p.W = W; p.eps = eps; p.eps_in = eps_in; p.eps_sum = eps_sum;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.out = out; p.pre_clamp = pre_clamp; p.summary = summary;
    return launch_forward(plan, grid, p, false, noisy, stream);
}

int bnn_feature_nn_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* W, const float* eps_in, uint64_t philox_seed,
                       int64_t draw_id0, int64_t system_id0, float* latents, void* stream) {
    if (!plan || !grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (grid->B == 0 || grid->J == 0) return 0;
    if (!W || !latents) return fail(BNN_ERR_INVALID, "W/latents is NULL");
    // the generic engine with its latents output switched on and no (mu, std) output: the pool and the tail run on in-kernel normals and
    // their results are dropped (the latents do not depend on them)
    bnn_grid g = *grid;
    g.engine = 1;
    FwdParams p{};
    p.x = x; p.W = W; p.eps_in = eps_in;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.latents = latents;
    return launch_forward(plan, &g, p, false, eps_in != nullptr || grid->noisy, stream);
}

int bnn_forward_lowp_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* W, const float* eps, uint64_t philox_seed,
                         int64_t draw_id0, int64_t system_id0, int32_t precision, float* out, float* pre_clamp, float* summary, void* stream) {
    if (!plan || !grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (precision < BNN_PREC_BF16 || precision > BNN_PREC_F16X3) return fail(BNN_ERR_INVALID, "precision must be one of BNN_PREC_BF16 .. BNN_PREC_F16X3");
    if (!plan->v50net || plan->tab[0].kin4 != 31) return fail(BNN_ERR_UNSUPPORTED, "the reduced-precision kernels are built for the pretrained network with the v50 column mask only");
    if (grid->noisy) return fail(BNN_ERR_UNSUPPORTED, "the reduced-precision kernels have no noisy form");
    if (grid->B == 0 || grid->J == 0) return 0;
    if (!W) return fail(BNN_ERR_INVALID, "W is NULL");
    FwdParams p{};
    p.x = x; p.W = W; p.eps = eps;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.out = out; p.pre_clamp = pre_clamp; p.summary = summary;
    return launch_forward(plan, grid, p, false, false, stream, precision);
}

int bnn_multiswag_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                      const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, const float* z1, const float* z2,
                      const float* eps, float scale, uint64_t philox_seed, int64_t draw_id0, int64_t system_id0,
                      float* W_workspace, float* out, float* pre_clamp, float* summary, void* stream) {
    if (!plan || !grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (grid->B == 0 || grid->J == 0) return 0;  // nothing to do (empty tensors have NULL data pointers)
    if (!w_avg || !w2_avg || !pre_D || !seed_idx) return fail(BNN_ERR_INVALID, "NULL ensemble argument");
    if (W_workspace) {  // sample every draw once, then the forward kernel reads the materialised vectors
        if (!grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
        int rc = bnn_swag_draw_f32(plan, w_avg, w2_avg, pre_D, S, K, seed_idx, grid->J, z1, z2, scale, philox_seed, draw_id0,
                                   W_workspace, stream);
        if (rc) return rc;
        return bnn_forward_f32(plan, grid, x, W_workspace, eps, nullptr, nullptr, philox_seed, draw_id0, system_id0, out, pre_clamp,
                               summary, stream);
    }
    if ((z1 == nullptr) != (z2 == nullptr)) return fail(BNN_ERR_INVALID, "z1 and z2 must both be given or both be NULL");
    if (S < 1) return fail(BNN_ERR_INVALID, "bad S");
    FwdParams p{};
    int rc = draw_consts(K, scale, &p.c1, &p.c2, MAXK);
    if (rc) return rc;
    p.scale = scale; p.K = K; p.S = S;
    p.x = x; p.w_avg = w_avg; p.w2_avg = w2_avg; p.pre_D = pre_D; p.seed_idx = seed_idx; p.z1 = z1; p.z2 = z2; p.eps = eps;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.out = out; p.pre_clamp = pre_clamp; p.summary = summary;
    return launch_forward(plan, grid, p, true, false, stream);
}

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

int bnn_regress_f32(const bnn_plan* pl, const float* summary, const float* W, int64_t J, int64_t B, float* out, float* pre_clamp,
                    void* stream) {
    if (!pl) return fail(BNN_ERR_INVALID, "plan is NULL");
    if (J < 0 || B < 0) return fail(BNN_ERR_INVALID, "bad J/B");
    if (J == 0 || B == 0) return 0;
    if (!summary || !W || !out) return fail(BNN_ERR_INVALID, "NULL argument");
    if ((B + 127) / 128 > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many systems for one launch");
    if (!pl->v50net) {   // the hparam-built network: natural accumulation order, the generic forward kernel's tail bit for bit
        const GenArch& g = pl->gen;
        GenRegressParams q{};
        q.B = B; q.std_lo = pl->arch.lowest_std; q.std_span = (float)(6.0 - (double)pl->arch.lowest_std);
        q.d = g.d; q.SM = g.SM; q.n_reg = g.n_reg;
        int ld = g.SM;
        for (int l = 0; l < g.n_reg; ++l) { q.layer[l] = g.layer[g.n_feat + l]; ld = std::max(ld, q.layer[l].N); }
        q.ld = ld | 1;   // odd: conflict-free per-thread rows
        for (int64_t j0 = 0; j0 < J; j0 += 65535) {
            const int64_t nj = J - j0 < 65535 ? J - j0 : 65535;
            q.summary = summary + j0 * B * g.SM; q.W = W + j0 * g.d; q.out = out + j0 * B * 2;
            q.pre = pre_clamp ? pre_clamp + j0 * B * 2 : nullptr;
            hipLaunchKernelGGL(bnn_regress_generic_kernel, dim3((unsigned)((B + 63) / 64), (unsigned)nj), dim3(64), (size_t)2 * 64 * q.ld * sizeof(float),
                               (hipStream_t)stream, q);
            HIP_TRY(hipGetLastError());
        }
        return 0;
    }
    RegressParams p;
    p.summary = summary; p.W = W; p.out = out; p.pre = pre_clamp; p.B = B;
    p.std_lo = pl->arch.lowest_std; p.std_span = (float)(6.0 - (double)pl->arch.lowest_std);
    const int SM = layout_of(pl->megno).SM;
    for (int l = 0; l < 3; ++l) {
        const std::vector<int32_t>& o = pl->tab[0].order[3 + l];
        if ((int)o.size() != (l == 0 ? SM : H)) return fail(BNN_ERR_INVALID, "internal: regress_nn order table");
        for (int i = 0; i < (int)o.size(); ++i) p.ord[l][i] = (int8_t)o[i];
    }
    for (int64_t j0 = 0; j0 < J; j0 += 65535) {  // grid.y limit
        const int64_t nj = J - j0 < 65535 ? J - j0 : 65535;
        RegressParams q = p;
        q.summary = summary + j0 * B * SM; q.W = W + j0 * pl->d; q.out = out + j0 * B * 2;
        q.pre = pre_clamp ? pre_clamp + j0 * B * 2 : nullptr;
        if (pl->megno) hipLaunchKernelGGL(bnn_regress_kernel<true>, dim3((unsigned)((B + 127) / 128), (unsigned)nj), dim3(128), 0, (hipStream_t)stream, q);
        else hipLaunchKernelGGL(bnn_regress_kernel<false>, dim3((unsigned)((B + 127) / 128), (unsigned)nj), dim3(128), 0, (hipStream_t)stream, q);
        HIP_TRY(hipGetLastError());
    }
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

int bnn_feature_pack_f64(const double* tseries, const double* mass, const double* X64_in, int64_t N, int32_t T, const double* mean,
                         const double* scale, double* X64_out, float* x32_out, void* stream) {
    if (N < 0 || T < 1) return fail(BNN_ERR_INVALID, "bad N/T");
    if (N == 0) return 0;
    if (!tseries && !X64_in) return fail(BNN_ERR_INVALID, "need tseries (+mass) or X64_in");
    if (tseries && !mass) return fail(BNN_ERR_INVALID, "tseries needs mass");
    if (!X64_out && !x32_out) return fail(BNN_ERR_INVALID, "no output requested");
    if (x32_out && (!mean || !scale)) return fail(BNN_ERR_INVALID, "x32_out needs mean and scale");
    const int64_t rows = N * T;
    if (tseries) {
        const int64_t nblk = (rows + PACK_ROWS - 1) / PACK_ROWS;
        if (nblk > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many rows for one launch");
        hipLaunchKernelGGL(bnn_feature_pack_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, tseries, mass, N, (int)T, mean, scale,
                           X64_out, x32_out);
    } else {
        const int64_t total = rows * F;
        if ((total + 255) / 256 > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many rows for one launch");
        hipLaunchKernelGGL(bnn_standardise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X64_in, total, mean,
                           scale, X64_out, x32_out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_moments_f64(const float* samples, int64_t R, int64_t B, double* moments, int32_t accumulate, void* stream) {
    if (R < 0 || B < 0) return fail(BNN_ERR_INVALID, "bad argument");
    if (B == 0) return 0;  // an empty shard (more ranks than systems) has NULL data pointers
    if (!samples || !moments) return fail(BNN_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(bnn_moments_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64, 16), 0, (hipStream_t)stream, samples, R, B, moments,
                       accumulate);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_philox_normal_f32(int32_t kind, uint64_t philox_seed, int64_t id0, int64_t n_rows, int64_t B, int64_t system_id0, int32_t width,
                          int32_t n_features, float* out, void* stream) {
    if (!out || kind < 0 || kind > 6 || n_rows < 0 || width < 0) return fail(BNN_ERR_INVALID, "bad argument");
    if (kind == 3 && n_features != 0 && n_features != F && n_features != 2 * F) return fail(BNN_ERR_INVALID, "n_features must be 41 or 82");
    const int NF = n_features > 0 ? n_features : F;
    int64_t total = kind == 2 ? n_rows * B * 2 * (int64_t)(width > 0 ? width : L) : kind == 4 ? n_rows * B * (int64_t)(width > 0 ? width : S2)
                    : kind == 3 ? n_rows * B * (int64_t)width * NF : kind == 5 ? n_rows * B * (int64_t)width : kind == 6 ? n_rows * B : n_rows * (int64_t)width;
    if (total == 0) return 0;
    if ((total + 255) / 256 > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "too many normals for one launch");
    hipLaunchKernelGGL(bnn_philox_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kind, philox_seed,
                       id0, n_rows, B, system_id0, width, NF, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_philox_raw_u32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int64_t n, uint32_t* out,
                       void* stream) {
    if (!out || n < 0) return fail(BNN_ERR_INVALID, "bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(bnn_philox_raw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c0, c1, c2, c3, k0, k1,
                       n, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- streaming statistics epilogue ----------------------------------------------------------------------------------
static int stats_params(const bnn_stats* st, StatsParams* sp) {
    if (!st) return fail(BNN_ERR_INVALID, "stats is NULL");
    if (st->tn_nsamp < 1 || st->tn_nsamp > 4096) return fail(BNN_ERR_RANGE, "tn_nsamp must be in [1, 4096]");
    const bool prior = st->prior_thr < INFINITY;
    if (prior && (!st->prior_surv || st->prior_m < 2 || !(st->prior_step > 0.0f))) return fail(BNN_ERR_INVALID, "prior table missing");
    sp->tn_nsamp = st->tn_nsamp; sp->tn_left = st->tn_left; sp->prior_thr = st->prior_thr;
    sp->prior_surv = st->prior_surv; sp->prior_m = st->prior_m; sp->prior_step = st->prior_step;
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

int bnn_multiswag_stats_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                            const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, const float* z1, const float* z2,
                            const float* eps, float scale, uint64_t philox_seed, int64_t draw_id0, int64_t system_id0,
                            float* W_workspace, const bnn_stats* st, float* t_out, void* stream) {
    if (!plan || !grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    StatsParams sp;
    int rc = stats_params(st, &sp);
    if (rc) return rc;
    if (grid->B == 0 || grid->J == 0) return 0;
    if (!w_avg || !w2_avg || !pre_D || !seed_idx || !W_workspace || !t_out) return fail(BNN_ERR_INVALID, "NULL argument (the statistics form needs a draw workspace)");
    rc = bnn_swag_draw_f32(plan, w_avg, w2_avg, pre_D, S, K, seed_idx, grid->J, z1, z2, scale, philox_seed, draw_id0, W_workspace, stream);
    if (rc) return rc;
    FwdParams p{};
    p.x = x; p.W = W_workspace; p.eps = eps;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.sink = t_out; p.st = sp;
    return launch_forward(plan, grid, p, false, false, stream);
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

// ---- slab drivers: the whole (systems x draws) grid reduced on the fly, nothing of size J x B ever in memory --------------------
// Both evaluate the draws in slabs of `draws_per_launch` (a multiple of nchunks) through caller-provided scratch, on `stream`,
// without synchronising: moments -> [B,4] float64; bands -> the quantile sketch (hist, mom) of the post-epilogue times.
static int slab_args(const bnn_plan* plan, const bnn_grid* grid, int32_t draws_per_launch) {
    if (!plan || !grid) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (grid->nchunks < 1 || grid->J < 0 || grid->J % grid->nchunks) return fail(BNN_ERR_INVALID, "J must be a multiple of nchunks");
    if (draws_per_launch < grid->nchunks || draws_per_launch % grid->nchunks) return fail(BNN_ERR_INVALID, "draws_per_launch must be a positive multiple of nchunks");
    return 0;
}

int bnn_multiswag_moments_f64(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                              const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, float scale, uint64_t philox_seed,
                              int64_t draw_id0, int64_t system_id0, int32_t draws_per_launch, float* W_workspace, float* out_workspace,
                              double* moments, void* stream) {
    int rc = slab_args(plan, grid, draws_per_launch);
    if (rc) return rc;
    if (grid->B == 0) return 0;
    if (!moments || !out_workspace || !W_workspace) return fail(BNN_ERR_INVALID, "NULL workspace / moments");
    // (the first slab overwrites, the others accumulate: no memset node -- see bnn_nonfinite.hip on memset nodes in captured graphs)
    if (grid->J == 0) HIP_TRY(hipMemsetAsync(moments, 0, sizeof(double) * 4 * (size_t)grid->B, (hipStream_t)stream));
    for (int32_t j0 = 0; j0 < grid->J; j0 += draws_per_launch) {
        bnn_grid g = *grid;
        g.J = grid->J - j0 < draws_per_launch ? grid->J - j0 : draws_per_launch;
        rc = bnn_multiswag_f32(plan, &g, x, w_avg, w2_avg, pre_D, S, K, seed_idx + j0, nullptr, nullptr, nullptr, scale, philox_seed,
                               draw_id0 + j0, system_id0, W_workspace, out_workspace, nullptr, nullptr, stream);
        if (rc) return rc;
        rc = bnn_moments_f64(out_workspace, g.J / g.nchunks, g.B, moments, j0 > 0 ? 1 : 0, stream);
        if (rc) return rc;
    }
    return 0;
}

int bnn_multiswag_bands_f32(const bnn_plan* plan, const bnn_grid* grid, const float* x, const float* w_avg, const float* w2_avg,
                            const float* pre_D, int32_t S, int32_t K, const int32_t* seed_idx, float scale, uint64_t philox_seed,
                            int64_t draw_id0, int64_t system_id0, int32_t draws_per_launch, float* W_workspace, float* t_workspace,
                            const bnn_stats* st, int32_t group, const bnn_sketch* sk, uint32_t* hist, double* mom, void* stream) {
    int rc = slab_args(plan, grid, draws_per_launch);
    if (rc) return rc;
    if (grid->B == 0) return 0;
    if (!t_workspace || !W_workspace || !hist || !mom) return fail(BNN_ERR_INVALID, "NULL workspace / sketch");
    for (int32_t j0 = 0; j0 < grid->J; j0 += draws_per_launch) {
        bnn_grid g = *grid;
        g.J = grid->J - j0 < draws_per_launch ? grid->J - j0 : draws_per_launch;
        rc = bnn_multiswag_stats_f32(plan, &g, x, w_avg, w2_avg, pre_D, S, K, seed_idx + j0, nullptr, nullptr, nullptr, scale,
                                     philox_seed, draw_id0 + j0, system_id0, W_workspace, st, t_workspace, stream);
        if (rc) return rc;
        rc = bnn_sketch_update_u32(t_workspace, g.J / g.nchunks, g.B, group, sk, hist, mom, stream);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"

