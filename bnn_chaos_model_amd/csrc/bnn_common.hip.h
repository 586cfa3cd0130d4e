// bnn_common.hip.h -- shared device code of the MultiSWAG kernels: vector types, build switches, Philox4x32-10
// normals, kernel parameters, MFMA/ReLU helpers, and the SWAG draw (SWAGModel.sample_weights,
// spock_reg_model.py:815-838) as a kernel and as a workgroup-prologue routine.  Included by bnn_kernels.hip only.
#pragma once
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

#define DEVINL __device__ __forceinline__

#ifndef BNN_PRIO_STAGGER
#define BNN_PRIO_STAGGER 0
#endif
#ifndef BNN_STAMPS
#define BNN_STAMPS 0  // diagnostic build: wave 0 of each workgroup of the 4x4x1 kernel sums s_memtime deltas per phase
#endif                // into the pre_clamp buffer (as uint64 [block][12]); never enabled in the shipped library
#if BNN_STAMPS
#define STAMP(i)                                                                 \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        unsigned long long _t;                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); \
        st_acc[i] += _t - st_prev;                                               \
        st_prev = _t;                                                            \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
#ifndef BNN_EXP
#define BNN_EXP 0  // timing experiments (wrong results when non-zero)
#endif
#ifndef BNN_WAVES_PER_SIMD
#define BNN_WAVES_PER_SIMD 3  // register budget of the 16x16x4 kernel: 2 -> 256 VGPRs, 3 -> 168 (+3 % measured)
#endif

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) and the normals derived from it.
// Counters use GLOBAL draw / output-row / system ids, so results are invariant to sharding.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t TAG_Z1 = 0x10000000u, TAG_Z2 = 0x20000000u, TAG_EPS = 0x30000000u, TAG_IN = 0x40000000u, TAG_SUM = 0x50000000u;

DEVINL uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a mul_hi + mul_lo pair: both are quarter-rate
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// Box-Muller on 24-bit uniforms in (0,1); v_sin/v_cos take revolutions, so no range reduction; v_log / v_sqrt as they
// are (1 ulp; the radicand lies in [1e-7, 34]): these normals are noise, and every consumer -- in-kernel or through
// bnn_philox_normal_f32 -- gets them from this one function.
DEVINL f32x2 box_muller(uint32_t a, uint32_t b) {
    float u1 = ((float)(a >> 8) + 0.5f) * 5.9604644775390625e-8f;
    float u2 = ((float)(b >> 8) + 0.5f) * 5.9604644775390625e-8f;
    float r = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1));
    f32x2 o;
    o.x = r * __builtin_amdgcn_cosf(u2);
    o.y = r * __builtin_amdgcn_sinf(u2);
    return o;
}

DEVINL f32x4 philox_normal4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed) {
    uint4 r = philox4x32_10(make_uint4(c0, c1, c2, c3), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    f32x2 a = box_muller(r.x, r.y), b = box_muller(r.z, r.w);
    f32x4 o = {a.x, a.y, b.x, b.y};
    return o;
}

// z1[draw][i], z2[draw][k]: counter = (tag | quad, draw lo, draw hi, 0)
DEVINL float philox_z(uint32_t tag, int64_t draw, int elem, uint64_t seed) {
    f32x4 n = philox_normal4(tag | (uint32_t)(elem >> 2), (uint32_t)draw, (uint32_t)((uint64_t)draw >> 32), 0u, seed);
    return n[elem & 3];
}
// eps[row][sys][kind][n], quad = (kind*20 + n) / 4: counter = (tag | quad, sys lo, sys hi16 | row hi16 << 16, row lo)
DEVINL f32x4 philox_sys4(uint32_t tag, int64_t row, int64_t sys, int quad, uint64_t seed) {
    uint32_t c2 = (uint32_t)(((uint64_t)sys >> 32) & 0xffffu) | ((uint32_t)(((uint64_t)row >> 32) & 0xffffu) << 16);
    return philox_normal4(tag | (uint32_t)quad, (uint32_t)sys, c2, (uint32_t)row, seed);
}
DEVINL f32x4 philox_eps4(int64_t row, int64_t sys, int quad, uint64_t seed) { return philox_sys4(TAG_EPS, row, sys, quad, seed); }
// input noise eps_in[row][sys][t][col] (:445): quad = t*11 + col/4 (rows padded to 44 so quads align with 4-column groups);
// summary noise eps_sum[row][sys][n] (:449): quad = n/4.

// ------------------------------------------------------------------------------------------------
// kernel parameters
// ------------------------------------------------------------------------------------------------
struct FwdParams {
    const float* x;
    int64_t B;
    int32_t T, ntiles;
    int32_t J, nch;
    int64_t csz;
    int32_t spc;  // systems per workgroup (multiple of 64)
    int32_t K, S;
    const float* W;  // [J,d] materialised draws (unfused) or nullptr
    const float* w_avg;
    const float* w2_avg;
    const float* pre_D;
    const int32_t* seed_idx;
    const float* z1;
    const float* z2;
    float c1, c2, scale;
    const float* eps;
    const float* eps_in;
    const float* eps_sum;
    uint64_t seed;
    int64_t draw_id0, row_id0, sys_id0;
    float* out;
    float* pre_clamp;
    float* summary;
    const int16_t* tab_f1;
    const int16_t* tab_f2;
    const int16_t* tab_f4;  // 4x4x1 image gather table (v50 mask) or nullptr
    const float* rcp_tab;  // [i] = 1/(i+1), correctly rounded
    uint64_t zero_mask;
    float std_lo, std_span;
};

DEVINL f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// nn.ReLU as ONE integer max on the bit pattern: negative floats (and -0.0) are negative ints -> +0.0.
DEVINL float relu1(float v) {
    int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// registers 2,3 of m-tile 2 are padding (nmap_hidden) and never consumed: NLIVE = 2 there
template <int NLIVE = 4>
DEVINL f32x4 relu4(f32x4 v) {
    f32x4 o = v;
#pragma unroll
    for (int i = 0; i < NLIVE; ++i) o[i] = relu1(v[i]);
    return o;
}

// ------------------------------------------------------------------------------------------------
// SWAG draw of rows [i0, i0+64) by one wave (SWAGModel.sample_weights, spock_reg_model.py:815-838).
// pre_D rows are staged through a wave-private LDS slab so the HBM/L2 read is one contiguous
// 64*K-float run; lane l then owns row i0+l and accumulates its K-term dot product in k order.
// Callers bracket the two phases with workgroup barriers (stage -> barrier -> compute -> barrier).
// ------------------------------------------------------------------------------------------------
DEVINL void draw_stage(const float* __restrict__ pre_D_s, int i0, int K, int lane, float* slab) {
    const int64_t base = (int64_t)i0 * K, lim = (int64_t)D * K;
    for (int n = 0; n < K; ++n) {
        int idx = n * 64 + lane;
        if (base + idx < lim) slab[idx] = pre_D_s[base + idx];
    }
}

DEVINL float draw_row(const float* __restrict__ w_avg_s, const float* __restrict__ w2_avg_s, int i, int K, int lane,
                      const float* slab, const float* zsh, float z1v, float c1, float c2, float scale) {
    // D = pre_D - w_avg[:,None] (:826); sigma = abs(diag(w2_avg - w_avg**2)) (:832)
    // w = w_avg + scale/sqrt2 * z1 @ sigma**0.5 (:834);  w += scale * (D @ z2).T / sqrt(2(K-1)) (:835)
    float wa = w_avg_s[i], w2 = w2_avg_s[i];
    float sq = wa * wa;
    float var = w2 - sq;
    float sd = sqrtf(fabsf(var));
    float t1 = (c1 * z1v) * sd;
    float w = wa + t1;
    float dot = 0.0f;
    const float* row = slab + lane * K;
    for (int k = 0; k < K; ++k) {
        float Dk = row[k] - wa;
        dot = fmaf(Dk, zsh[k], dot);
    }
    float t2 = (scale * dot) / c2;
    return w + t2;
}

constexpr int SLAB = 64 * MAXK;  // floats per wave

// Slab-free variant for the single-launch prologue: thread-per-element, the K-term row read straight from L2.
// Same operation sequence as draw_row, hence the same bits.
DEVINL float draw_row_direct(const float* __restrict__ w_avg_s, const float* __restrict__ w2_avg_s,
                             const float* __restrict__ pre_D_s, int i, int K, const float* zsh, float z1v, float c1, float c2,
                             float scale) {
    float wa = w_avg_s[i], w2 = w2_avg_s[i];
    float sq = wa * wa;
    float var = w2 - sq;
    float sd = sqrtf(fabsf(var));
    float t1 = (c1 * z1v) * sd;
    float w = wa + t1;
    float dot = 0.0f;
    const float* row = pre_D_s + (int64_t)i * K;
#pragma unroll 6
    for (int k = 0; k < K; ++k) {
        float Dk = row[k] - wa;
        dot = fmaf(Dk, zsh[k], dot);
    }
    float t2 = (scale * dot) / c2;
    return w + t2;
}

__global__ __launch_bounds__(256) void bnn_swag_draw_kernel(const float* __restrict__ w_avg, const float* __restrict__ w2_avg,
                                                            const float* __restrict__ pre_D, int S, int K,
                                                            const int32_t* __restrict__ seed_idx, const float* __restrict__ z1,
                                                            const float* __restrict__ z2, float c1, float c2, float scale,
                                                            uint64_t seed, int64_t draw_id0, float* __restrict__ W_out) {
    __shared__ float slabs[4 * SLAB];
    __shared__ float zsh[MAXK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.y;
    int s = seed_idx[e];
    const bool bad = (s < 0 || s >= S);
    if (bad) s = 0;
    if (threadIdx.x < K)
        zsh[threadIdx.x] = z2 ? z2[(int64_t)e * K + threadIdx.x] : philox_z(TAG_Z2, draw_id0 + e, threadIdx.x, seed);
    const int i0 = (blockIdx.x * 4 + wave) * 64;
    const float* pd = pre_D + (int64_t)s * D * K;
    if (i0 < D) draw_stage(pd, i0, K, lane, slabs + wave * SLAB);
    __syncthreads();
    const int i = i0 + lane;
    if (i < D) {
        float z1v = z1 ? z1[(int64_t)e * D + i] : philox_z(TAG_Z1, draw_id0 + e, i, seed);
        float w = draw_row(w_avg + (int64_t)s * D, w2_avg + (int64_t)s * D, i, K, lane, slabs + wave * SLAB, zsh, z1v, c1, c2,
                           scale);
        W_out[(int64_t)e * D + i] = bad ? __builtin_nanf("") : w;
    }
}

