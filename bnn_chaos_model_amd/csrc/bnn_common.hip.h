// bnn_common.hip.h -- device code shared by the kernel translation units: vector types, Philox4x32-10 and the normals
// derived from it, MFMA / ReLU helpers, and the SWAG draw (SWAGModel.sample_weights, spock_reg_model.py:815-838) as a
// per-row routine used by the draw kernel and by the forward kernel's in-prologue draw.
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

#include "bnn_internal.h"

namespace bnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

#define DEVINL __device__ __forceinline__

constexpr int SCR4 = 2 * 16 * S2;  // floats of LDS scratch per wave of a forward kernel: Philox normals + summaries of 16 systems

// ------------------------------------------------------------------------------------------------
// Work item of a forward workgroup: (draw e, block `sub` of the chunk of systems that draw covers).
// The grid holds nsub * J workgroups.  Work order w: output row r = w % R fastest (R = J / nch draws share a chunk), then the
// block of systems, then the chunk -- so consecutive w are the SAME systems under different draws.  Workgroups are dealt to
// the 8 XCDs round-robin by block id (MI355X_MICROARCH.md; a speed assumption only, never correctness), and each XCD has its own
// 4 MiB L2, so XCD k takes the k-th CONTIGUOUS eighth of the work order: the workgroups resident on one XCD then stream the same
// few hundred systems through that XCD's L2 together, and x comes from HBM once.  (With the plain order id -> (draw, block) every
// XCD touched every active block of systems; at nchunks = 10 that was ten 8 MB regions per 4 MiB L2, all of the x traffic fell
// through to the Infinity Cache and the bf16 kernels sat at its 8 TB/s.)  The host switches it on (p.xcd_order) when draws are
// chunked or x is larger than the Infinity Cache; for a dense grid over an x that the Infinity Cache holds (configs[1]: 164 MB)
// the plain order measured 1.4 % faster (same-box A/B), at configs[2] (16.4 GB) the XCD order 2.2 % faster.
// ------------------------------------------------------------------------------------------------
struct WorkItem {
    int e;        // draw
    int64_t sub;  // block of systems within the draw's chunk
};
DEVINL WorkItem work_item(const FwdParams& p) {
    const int64_t nblk = gridDim.x, b = blockIdx.x;
    const int64_t per = nblk >> 3;
    const int64_t w = (p.xcd_order && b < (per << 3)) ? (b & 7) * per + (b >> 3) : b;
    const int R = p.J / p.nch;
    const int64_t nsub = nblk / p.J;
    const int r = (int)(w % R);
    const int64_t t = w / R;
    WorkItem wi;
    wi.sub = t % nsub;
    wi.e = r * p.nch + (int)(t / nsub);
    return wi;
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) and the normals derived from it.
// Counters use GLOBAL draw / output-row / system ids, so results are invariant to sharding.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t TAG_Z1 = 0x10000000u, TAG_Z2 = 0x20000000u, TAG_EPS = 0x30000000u, TAG_IN = 0x40000000u, TAG_SUM = 0x50000000u,
                   TAG_TN = 0x60000000u, TAG_U = 0x70000000u, TAG_TNS = 0x80000000u, TAG_US = 0x90000000u;

template <int ROUNDS>
DEVINL uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a mul_hi + mul_lo pair: both are quarter-rate
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // three-input xor in ONE instruction (v_bitop3_b32, truth table 0x96): the compiler does not form it by itself
        c = make_uint4(__builtin_amdgcn_bitop3_b32(hi1, c.y, k.x, 0x96), lo1, __builtin_amdgcn_bitop3_b32(hi0, c.w, k.y, 0x96), lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
DEVINL uint4 philox4x32_10(uint4 c, uint2 k) { return philox4x32<10>(c, k); }

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
// per-(output row, system) streams: counter = (tag | quad, sys lo, sys hi16 | row hi16 << 16, row lo)
DEVINL uint4 philox_sys_ctr(uint32_t tag, int64_t row, int64_t sys, int quad) {
    uint32_t c2 = (uint32_t)(((uint64_t)sys >> 32) & 0xffffu) | ((uint32_t)(((uint64_t)row >> 32) & 0xffffu) << 16);
    return make_uint4(tag | (uint32_t)quad, (uint32_t)sys, c2, (uint32_t)row);
}
// eps[row][sys][kind][n], quad = (kind*20 + n) / 4
DEVINL f32x4 philox_sys4(uint32_t tag, int64_t row, int64_t sys, int quad, uint64_t seed) {
    uint4 c = philox_sys_ctr(tag, row, sys, quad);
    return philox_normal4(c.x, c.y, c.z, c.w, seed);
}
DEVINL f32x4 philox_eps4(int64_t row, int64_t sys, int quad, uint64_t seed) { return philox_sys4(TAG_EPS, row, sys, quad, seed); }
// summary noise eps_sum[row][sys][n] (:449): quad = n/4.

// The input noise of forward(noisy_val=True) (:445) is 4 100 normals per evaluation and, next to v_mfma_f32_4x4x1 (which holds
// the SIMD's vector issue port for its whole duration: profiles/r02_coexec2_probe.txt), every vector cycle spent on it is paid in
// full.  So this stream takes SIX normals from each Philox block instead of four: the 128 bits are cut into six 21-bit
// uniforms; a uniform becomes a float in [1, 2) with ONE shift/align + ONE and-or on the bit pattern (no int->float convert,
// 8 issue cycles on this chip): the angle is used as it stands (v_sin / v_cos take revolutions and are periodic), the radius
// level is 2 - f in (0, 1).  Normals reach 5.4 sigma; 2^21 distinct angles.
// This one stream (TAG_IN: 99 % of all random numbers of a noisy evaluation) runs Philox4x32 with SEVEN rounds, the smallest round
// count Salmon et al. (SC'11, table 2) report as passing BigCrush ("Crush-resistant"); every other stream keeps the customary ten.
// eps_in[row][sys][t][col]: block = t*7 + col/6, normal col%6 of the block.
constexpr int NIN_PER_BLOCK = 6, NIN_BLOCKS = 7;  // 7 blocks x 6 >= 41 columns
constexpr int NIN_ROUNDS = 7;
// The and-or is ONE v_and_or_b32 only if at most one of its two constants comes from the scalar side (gfx950 VALU instructions take
// one scalar / literal operand): the OR constant is therefore kept in a VGPR the compiler cannot see through -- left to itself it
// emits v_and_b32 + v_or_b32 with two literals, 42 extra vector instructions per 64-row tile of the noisy forward.
DEVINL uint32_t unit21_orc() {
    uint32_t c = 0x3F800002u;
    asm("" : "+v"(c));   // (not volatile: one copy per kernel is enough, the compiler may hoist and share it)
    return c;
}
DEVINL float unit21(uint32_t aligned, uint32_t orc) {  // bits [22:2] of `aligned` are the 21-bit field; +half a step so that f is never 1 or 2
    return __builtin_bit_cast(float, (aligned & 0x007FFFFCu) | orc);
}
DEVINL f32x2 box_muller21(float f_radius, float f_angle) {
    const float u1 = 2.0f - f_radius;
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln2 log2 u1)
    const f32x2 cs = {__builtin_amdgcn_cosf(f_angle), __builtin_amdgcn_sinf(f_angle)};
    return cs * (f32x2){r, r};   // one v_pk_mul_f32
}
DEVINL void philox_normal6(uint4 ctr, uint64_t seed, float (&n)[6]) {
    const uint4 r = philox4x32<NIN_ROUNDS>(ctr, make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const uint32_t orc = unit21_orc();
    // fields (bit offsets in the 128-bit block x:y:z:w, little end first): 0, 21, 42, 63, 84, 105 -- each moved to bits [22:2]
    const float f0 = unit21(r.x << 2, orc);
    const float f1 = unit21(__builtin_amdgcn_alignbit(r.y, r.x, 19), orc);
    const float f2 = unit21(r.y >> 8, orc);
    const float f3 = unit21(__builtin_amdgcn_alignbit(r.z, r.y, 29), orc);
    const float f4 = unit21(__builtin_amdgcn_alignbit(r.w, r.z, 18), orc);
    const float f5 = unit21(r.w >> 7, orc);
    const f32x2 a = box_muller21(f0, f1), b = box_muller21(f2, f3), c = box_muller21(f4, f5);
    n[0] = a.x; n[1] = a.y; n[2] = b.x; n[3] = b.y; n[4] = c.x; n[5] = c.y;
}
DEVINL void philox_in6(int64_t row, int64_t sys, int block, uint64_t seed, float (&n)[6]) {
    philox_normal6(philox_sys_ctr(TAG_IN, row, sys, block), seed, n);
}

// MEASUREMENT BUILDS ONLY (-DBNN_NIN16=1, round 6): EIGHT 16-bit uniforms per Philox block instead of six 21-bit
// ones -- six blocks per 41-column row instead of seven, the last block finishing ONE Box-Muller pair (column 40) instead of four.  Only
// the pretrained network's noisy kernel (in-kernel Philox form) switches; bnn_philox_normal_f32, the generic engine and the fix-up keep
// the six-per-block stream, so a build with this flag fails the explicit == in-kernel tests by construction.
#ifndef BNN_NIN16
#define BNN_NIN16 0
#endif
constexpr int NIN16_PER_BLOCK = 8, NIN16_BLOCKS = 6;
DEVINL uint32_t unit16_orc() {
    uint32_t c = 0x3F800040u;   // +half a step of the 16-bit field: f is never 1 or 2
    asm("" : "+v"(c));
    return c;
}
DEVINL float unit16(uint32_t aligned, uint32_t orc) {   // bits [22:7] of `aligned` are the 16-bit field
    return __builtin_bit_cast(float, (aligned & 0x007FFF80u) | orc);
}
template <int NPAIRS>
DEVINL void philox_in8(int64_t row, int64_t sys, int block, uint64_t seed, float (&n)[8]) {
    const uint4 r = philox4x32<NIN_ROUNDS>(philox_sys_ctr(TAG_IN, row, sys, block), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const uint32_t orc = unit16_orc();
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < NPAIRS) {
            const f32x2 a = box_muller21(unit16(w[q] << 7, orc), unit16(w[q] >> 9, orc));   // radius from the low half, angle from the high half
            n[2 * q] = a.x;
            n[2 * q + 1] = a.y;
        } else {
            n[2 * q] = n[2 * q + 1] = 0.0f;
        }
    }
}

DEVINL f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
DEVINL f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// v_mfma_f32_4x4x1_16b_f32 with CBSZ = 4: the A operand (4 neurons x 1 input) of block ABID serves all 16 blocks
// (scripts/probes/cbsz_probe.hip confirms the semantics on gfx950).  cbsz / abid are immediates: compile-time loops below.
template <int ABID>
DEVINL f32x4 mfma4b(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0); }

template <class Fn, int... I>
DEVINL void static_for_impl(Fn&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class Fn>
DEVINL void static_for(Fn&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// the same with an early exit: f returns false to stop (short-circuit: the remaining steps are branched over)
template <class Fn, int... I>
DEVINL void static_while_impl(Fn&& f, std::integer_sequence<int, I...>) { (void)(f(std::integral_constant<int, I>{}) && ...); }
template <int N, class Fn>
DEVINL void static_while(Fn&& f) { static_while_impl(f, std::make_integer_sequence<int, N>{}); }

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

// nn.ReLU with torch.relu's treatment of non-finite values: NaN is not <= 0 and stays NaN (whatever its sign bit -- the integer max
// above turns a NaN with the sign bit set into 0), -inf becomes 0, +inf stays.  Two instructions instead of one: used where the count
// does not matter and a non-finite value can arise from FINITE inputs -- regress_nn, behind a pool whose variance overflowed (the
// reference returns NaN there, tests/golden/make_golden_nonfinite.py system 19).  Finite values: the same bits as relu1.
DEVINL float relu_ieee(float v) { return v <= 0.0f ? 0.0f : v; }
template <int NLIVE = 4>
DEVINL f32x4 relu4_ieee(f32x4 v) {
    f32x4 o = v;
#pragma unroll
    for (int i = 0; i < NLIVE; ++i) o[i] = relu_ieee(v[i]);
    return o;
}

template <int CTRL>
DEVINL float quad_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// predict_instability's soft_clamp (:295-296, :437-442)
DEVINL f32x2 soft_clamp2(float r0, float r1, float std_lo, float std_span) {
    f32x2 o;
    o.x = (0.5f * (tanhf(r0) + 1.0f)) * 8.0f + 4.0f;
    o.y = (0.5f * (tanhf(r1) + 1.0f)) * std_span + std_lo;
    return o;
}

// compute_summary_stats for one latent (:420-431), op for op: torch.std (unbiased) from the pooled M2, **2, the two standard errors,
// the two sampled moments, sqrt(abs(.) + EPSILON).  One routine for the three forward kernels (fp32, reduced precision, generic).
// BNN_DIVFAST (A/B builds ONLY; profiles/r04_ab_variants.txt): the three divisions by the wave-uniform constants T - 1 and T as the
// 3-instruction sequence q = x r, e = fma(-q, n, x), q + e r with r = RN(1 / n) instead of the IEEE divide expansion -- to MEASURE what
// the lever is worth (round-3 verdict, item 5a).  Not the product: the sequence is correctly rounded for most but not provably all
// operands, and the parity path divides as the oracle does.
#ifndef BNN_DIVFAST
#define BNN_DIVFAST 0
#endif
DEVINL float div_by(float x, float n) {
#if BNN_DIVFAST
    const float rn = 1.0f / n;                 // wave-uniform: hoisted out of every loop by the compiler
    const float q = x * rn;
    const float e = fmaf(-q, n, x);
    return fmaf(e, rn, q);
#else
    return x / n;
#endif
}
DEVINL void sampled_moments(float sample_mu, float m2sum, float e1, float e2, float nm1, float nT, float& mu_s, float& sd_s) {
    const float sd = sqrtf(div_by(m2sum, nm1));  // torch.std (unbiased)
    const float sample_var = sd * sd;            // **2
    const float std_in_mu = sqrtf(div_by(sample_var, nT));
    const float std_in_var = sqrtf(div_by(2.0f * (sample_var * sample_var), nm1));
    mu_s = e1 * std_in_mu + sample_mu;
    const float var_s = e2 * std_in_var + sample_var;
    sd_s = sqrtf(fabsf(var_s) + 1e-5f);        // EPSILON (:337)
}

// regress_nn of the pretrained network (40 (42) -> 40 -> 40 -> 2) for the 16 systems of a wave-batch on v_mfma_f32_16x16x4_f32 (exact
// fp32): column c = lane & 15 <-> system, lane group g = lane >> 4 <-> k within a k-step.  skeep[ks] = this lane's B operand of k-step ks
// of regress_nn.0 (kmap_summary; summary noise already added), f2frag = the gathered operand fragments [Lay::NF2][64] in LDS
// (bnn_tables.cpp).  Returns the accumulator of regress_nn.4: [0], [1] = the two pre-clamp outputs in lanes g = 0.
template <bool MEGNO>
DEVINL f32x4 regress16(const float (&skeep)[Lay<MEGNO>::NK4], const float* f2frag, int lane) {
    using Y = Lay<MEGNO>;
    const float* f2l = f2frag + lane;
    auto W2f = [&](int f) { return f2l[f * 64]; };
    f32x4 a4[3], a5[3], a6;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
        a4[mt] = (f32x4){W2f(Y::F_B4 + mt * 4), W2f(Y::F_B4 + 1 + mt * 4), W2f(Y::F_B4 + 2 + mt * 4), W2f(Y::F_B4 + 3 + mt * 4)};
#pragma unroll
    for (int ks = 0; ks < Y::NK4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) a4[mt] = mfma(W2f(Y::F_L4 + ks * 3 + mt), skeep[ks], a4[mt]);
    a4[0] = relu4_ieee(a4[0]); a4[1] = relu4_ieee(a4[1]); a4[2] = relu4_ieee<2>(a4[2]);
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
        a5[mt] = (f32x4){W2f(Y::F_B5 + mt * 4), W2f(Y::F_B5 + 1 + mt * 4), W2f(Y::F_B5 + 2 + mt * 4), W2f(Y::F_B5 + 3 + mt * 4)};
#pragma unroll
    for (int ks = 0; ks < NKH; ++ks)
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) a5[mt] = mfma(W2f(Y::F_L5 + ks * 3 + mt), a4[ks >> 2][ks & 3], a5[mt]);
    a5[0] = relu4_ieee(a5[0]); a5[1] = relu4_ieee(a5[1]); a5[2] = relu4_ieee<2>(a5[2]);
    a6 = (f32x4){W2f(Y::F_B6), W2f(Y::F_B6 + 1), W2f(Y::F_B6 + 2), W2f(Y::F_B6 + 3)};
#pragma unroll
    for (int ks = 0; ks < NKH; ++ks) a6 = mfma(W2f(Y::F_L6 + ks), a5[ks >> 2][ks & 3], a6);
    return a6;
}

// ------------------------------------------------------------------------------------------------
// SWAG draw of rows [i0, i0+64) by one wave (SWAGModel.sample_weights, spock_reg_model.py:815-838).
// pre_D rows are staged through a wave-private LDS slab, MAXK columns at a time (for K <= 32 the read is one contiguous
// 64*K-float run); lane l then owns row i0+l and accumulates its K-term dot product in k order.
// Callers bracket the two phases with workgroup barriers (stage -> barrier -> compute -> barrier), once per chunk of columns.
// ------------------------------------------------------------------------------------------------
DEVINL void draw_stage(const float* __restrict__ pre_D_s, int i0, int d, int K, int kc, int Kc, int lane, float* slab) {
    // columns [kc, kc + Kc) of rows [i0, i0 + 64): slab[r * Kc + c].  With kc = 0, Kc = K (every K <= 32) this is one contiguous run:
    // a plain copy, eight loads in flight per lane (the general loop below divides by Kc and waits for every load: 30 memory round trips in
    // a row, 15 us for ONE draw -- nothing next to a configs[2] step, a quarter of a 3 000-row call of the evaluation scripts).
    if (kc == 0 && Kc == K) {
        const int rows = d - i0 < 64 ? d - i0 : 64;
        const int total = rows * K;
        const float* src = pre_D_s + (int64_t)i0 * K;
        for (int n0 = 0; n0 < K; n0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = (n0 + u) * 64 + lane;
                v[u] = (n0 + u < K && idx < total) ? src[idx] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = (n0 + u) * 64 + lane;
                if (n0 + u < K && idx < total) slab[idx] = v[u];
            }
        }
        return;
    }
    for (int n = 0; n < Kc; ++n) {
        const int idx = n * 64 + lane, r = idx / Kc, c = idx - r * Kc;
        if (i0 + r < d) slab[idx] = pre_D_s[(int64_t)(i0 + r) * K + kc + c];
    }
}

// D = pre_D - w_avg[:,None] (:826); sigma = abs(diag(w2_avg - w_avg**2)) (:832)
// w = w_avg + scale/sqrt2 * z1 @ sigma**0.5 (:834);  w += scale * (D @ z2).T / sqrt(2(K-1)) (:835)
// The K-term dot product is accumulated in k order, a chunk of columns at a time (draw_dot), then draw_finish adds it.
DEVINL float draw_head(float wa, float w2, float z1v, float c1) {
    float sq = wa * wa;
    float var = w2 - sq;
    float sd = sqrtf(fabsf(var));
    float t1 = (c1 * z1v) * sd;
    return wa + t1;
}
DEVINL float draw_dot(const float* row, float wa, const float* zc, int Kc, float dot) {
    for (int k = 0; k < Kc; ++k) {
        float Dk = row[k] - wa;
        dot = fmaf(Dk, zc[k], dot);
    }
    return dot;
}
DEVINL float draw_finish(float w, float dot, float c2, float scale) {
    float t2 = (scale * dot) / c2;
    return w + t2;
}

constexpr int SLAB = 64 * MAXK;  // floats per wave
constexpr int MAXK_DRAW = 256;   // SWAG rank the draw kernel takes (the in-prologue draw of the v50 kernels: MAXK)

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
    if (K == 30) {
        // the reference's rank (run_swag.py:37): a row is 120 bytes -- seven 16-byte loads and one 8-byte load (global loads need dword
        // alignment only), ALL in flight before the first fmaf: one L2 round trip per element instead of five, and 8 instead of 30
        // requests per lane at the texture addresser, which is what bounds this loop (64 lanes x 120-byte stride = 60 cache lines per
        // instruction).  The k-ordered chain below is the same chain: same bits.
        f32x4 v[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) v[q] = *reinterpret_cast<const f32x4u*>(row + 4 * q);
        const f32x2 t = *reinterpret_cast<const f32x2u*>(row + 28);
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            dot = fmaf(v[q].x - wa, zsh[4 * q], dot);
            dot = fmaf(v[q].y - wa, zsh[4 * q + 1], dot);
            dot = fmaf(v[q].z - wa, zsh[4 * q + 2], dot);
            dot = fmaf(v[q].w - wa, zsh[4 * q + 3], dot);
        }
        dot = fmaf(t.x - wa, zsh[28], dot);
        dot = fmaf(t.y - wa, zsh[29], dot);
    } else {
#pragma unroll 6
        for (int k = 0; k < K; ++k) {
            float Dk = row[k] - wa;
            dot = fmaf(Dk, zsh[k], dot);
        }
    }
    float t2 = (scale * dot) / c2;
    return w + t2;
}

}  // namespace bnn
