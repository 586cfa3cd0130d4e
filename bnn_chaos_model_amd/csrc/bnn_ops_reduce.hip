// bnn_ops_reduce.hip -- predictive moments over the draws, and predict_instability on an explicit summary (spock_reg_model.py:437-442).
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
    GenLayer layer[GEN_MAX_LAYERS];   // regress_nn alone may hold all but one of the network's Linear modules (depth_in = 0: feature_nn is ONE Linear)
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

extern "C" {

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
        if (g.n_reg < 1 || g.n_reg > GEN_MAX_LAYERS) return fail(BNN_ERR_UNSUPPORTED, "internal: regress_nn has more Linear modules than the parameter block holds");
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

int bnn_moments_f64(const float* samples, int64_t R, int64_t B, double* moments, int32_t accumulate, void* stream) {
    if (R < 0 || B < 0) return fail(BNN_ERR_INVALID, "bad argument");
    if (B == 0) return 0;  // an empty shard (more ranks than systems) has NULL data pointers
    if (!samples || !moments) return fail(BNN_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(bnn_moments_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64, 16), 0, (hipStream_t)stream, samples, R, B, moments,
                       accumulate);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
