// bnn_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the MultiSWAG inference hot path and the
// C ABI of include/bnn_chaos_hip.h.
//
// Reference path (MilesCranmer/bnn_chaos_model): SWAGModel.sample_weights + forward_swag_fast /
// VarModel.forward in spock_reg_model.py:415-450, 486-528, 815-908, driven by
// figures/spock/regression.py:74-92 and figures/multiswag_5_planet.py:295-298.
//
// Kernel shape (DESIGN.md has the long form):
//   * one 256-thread workgroup = 4 independent waves; a workgroup serves ONE weight draw and a
//     block of systems.  Prologue: the draw (or a copy of a materialised draw) lands in LDS as the
//     reference's flat parameter vector; each wave gathers its MFMA A-operand fragments (74 VGPRs
//     for the v50 column mask) from LDS once and keeps them in registers.
//   * main loop, per wave, no barriers, no LDS: 16 rows (4 systems x 4 consecutive timesteps) per
//     step are read from HBM straight into the B-operand layout of v_mfma_f32_16x16x4_f32
//     (each lane: 8 consecutive floats of its row), then 24 + 30 + 20 MFMAs evaluate the three
//     feature_nn layers; accumulators of one layer are the B operands of the next (bnn_layout.h).
//   * time pooling: per-lane Welford over the lane's 25 timesteps, merged over the 4 lanes of a
//     quad (Chan), then the reference's sampled-moment formulas with explicit or Philox noise.
//   * regress_nn for 16 systems at a time on the same MFMA path, soft_clamp, 8-byte stores.
// fp32 MFMA is bit-for-bit a k-ordered fmaf chain, so the whole forward is reproducible on a CPU
// with the accumulation order exported by bnn_plan_layer_order().
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (explicit fmaf/MFMA are the only fusions).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bnn_chaos_hip.h"
#include "bnn_layout.h"
#include "bnn_tables.h"

using namespace bnn;

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
#ifndef BNN_TWO_STREAMS
#define BNN_TWO_STREAMS 0  // interleave two tiles per wave through the layers (0 = one tile at a time)
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
        uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// Box-Muller on 24-bit uniforms in (0,1); v_sin/v_cos take revolutions, so no range reduction.
DEVINL f32x2 box_muller(uint32_t a, uint32_t b) {
    float u1 = ((float)(a >> 8) + 0.5f) * 5.9604644775390625e-8f;
    float u2 = ((float)(b >> 8) + 0.5f) * 5.9604644775390625e-8f;
    float r = sqrtf(-2.0f * __logf(u1));
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

// ------------------------------------------------------------------------------------------------
// x tile -> B operands.  Lane (g, c) owns row `rowp` (41 floats) and k slots kmap_input(NK1, s, g).
// ------------------------------------------------------------------------------------------------
template <int NK1>
struct XTile {
    float v[NK1];
};
// Raw loaded registers.  The selects that build the B operands are applied at USE time (xtile()), never at
// load time: a select right after the load makes the compiler wait for the data there, which turns the
// one-tile-ahead prefetch into a stall of a full memory latency per tile.
template <int NK1>
struct XRaw {
    f32x4 a, b;
    f32x3 c;  // NK1 == 8: c.x = column 0
};

template <int NK1>
DEVINL XRaw<NK1> load_x(const float* __restrict__ rowp, int g) {
    XRaw<NK1> t;
    if constexpr (NK1 == 8) {
        const float* p = rowp + 8 + 8 * g;
        t.a = *reinterpret_cast<const f32x4u*>(p);
        t.b = *reinterpret_cast<const f32x4u*>(p + 4);  // group 3: columns 36..39, 38/39 replaced in xtile()
        t.c.x = rowp[0];
    } else {
        const float* p = rowp + 11 * g;
        t.a = *reinterpret_cast<const f32x4u*>(p);
        t.b = *reinterpret_cast<const f32x4u*>(p + 4);
        const float* pc = (g == 3) ? rowp + 38 : p + 8;  // group 3 has no columns 41..43: stay inside the row
        t.c = *reinterpret_cast<const f32x3u*>(pc);
    }
    return t;
}

template <int NK1>
DEVINL XTile<NK1> xtile(const XRaw<NK1>& r, int g) {
    XTile<NK1> t;
    t.v[0] = r.a.x; t.v[1] = r.a.y; t.v[2] = r.a.z; t.v[3] = r.a.w;
    t.v[4] = r.b.x; t.v[5] = r.b.y;
    if constexpr (NK1 == 8) {
        t.v[6] = (g == 3) ? r.c.x : r.b.z;   // slot (6, group 3) = column 0
        t.v[7] = (g == 3) ? 1.0f : r.b.w;    // slot (7, group 3) = bias
    } else {
        t.v[6] = r.b.z; t.v[7] = r.b.w;
        t.v[8] = (g == 3) ? 1.0f : r.c.x;    // slot 41 = bias
        t.v[9] = (g == 3) ? 0.0f : r.c.y;
        t.v[10] = (g == 3) ? 0.0f : r.c.z;
    }
    return t;
}

// ------------------------------------------------------------------------------------------------
// the fused kernel
// ------------------------------------------------------------------------------------------------
template <int NK1, bool NOISY, bool FUSED>
__global__ __launch_bounds__(256, (NOISY || NK1 != 8) ? 2 : BNN_WAVES_PER_SIMD) void bnn_multiswag_kernel(const FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* flat = lds;                 // [FLAT_LDS] flat parameter vector + zero slot
    float* zsh = lds + FLAT_LDS;       // [MAXK]
    float* f2frag = lds;               // [NF2][64] regress_nn operands in fragment order: OVERWRITES flat (below)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;

    // ---- work item: draw e, block `sub` of its chunk of systems (torch.chunk semantics)
    const int64_t id = blockIdx.x;
    const int e = (int)(id % p.J);
    const int64_t sub = id / p.J;
    const int ch = e % p.nch;
    const int64_t r = e / p.nch;  // output row
    const int64_t seg0 = (int64_t)ch * p.csz;
    const int64_t seg1 = (seg0 + p.csz < p.B) ? seg0 + p.csz : p.B;
    const int64_t b0 = seg0 + sub * p.spc;
    const int64_t b1 = (b0 + p.spc < seg1) ? b0 + p.spc : seg1;
    if (b0 >= b1) return;

    // ---- prologue: flat parameter vector of draw e -> LDS
    bool bad_seed = false;
    if constexpr (FUSED) {
        int s = p.seed_idx[e];
        bad_seed = (s < 0 || s >= p.S);
        if (bad_seed) s = 0;
        const int K = p.K;
        if (tid < K) zsh[tid] = p.z2 ? p.z2[(int64_t)e * K + tid] : philox_z(TAG_Z2, p.draw_id0 + e, tid, p.seed);
        const float* wa = p.w_avg + (int64_t)s * D;
        const float* w2 = p.w2_avg + (int64_t)s * D;
        const float* pd = p.pre_D + (int64_t)s * D * K;
        __syncthreads();  // zsh
        for (int i = tid; i < D; i += 256) {
            float z1v = p.z1 ? p.z1[(int64_t)e * D + i] : philox_z(TAG_Z1, p.draw_id0 + e, i, p.seed);
            flat[i] = draw_row_direct(wa, w2, pd, i, K, zsh, z1v, p.c1, p.c2, p.scale);
        }
    } else {
        const float* We = p.W + (int64_t)e * D;
        for (int i = tid; i < D; i += 256) flat[i] = We[i];
    }
    if (tid == 0) flat[ZERO_IDX] = 0.0f;
    __syncthreads();

    // ---- feature_nn operands: registers for the whole workgroup lifetime
    constexpr int NW1 = 3 * NK1, IW2 = NW1, IW3 = IW2 + 30, IB2 = IW3 + 20, IB3 = IB2 + 12, NF1 = IB3 + 8;
    float wf[NF1];
#pragma unroll
    for (int f = 0; f < NF1; ++f) wf[f] = flat[p.tab_f1[f * 64 + lane]];

    float in_scale[NOISY ? NK1 : 1], sum_scale[NOISY ? 10 : 1];
    if constexpr (NOISY) {
#pragma unroll
        for (int s = 0; s < NK1; ++s) {
            int col = 11 * g + s;
            in_scale[s] = col < F ? expf(flat[OFF_INLV + col] / 2.0f) : 0.0f;  // exp(logvar/2), :445
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) sum_scale[k] = expf(flat[OFF_SUMLV + kmap_summary(k, g)] / 2.0f);  // :449
    }
    // regress_nn operands -> LDS in fragment order (read back with immediate offsets, once per 16 systems).
    // They replace the flat vector in place: gather to registers, barrier, write.
    {
        constexpr int PER = (NF2 + 3) / 4;
        float tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int f = wave + 4 * i;
            tmp[i] = f < NF2 ? flat[p.tab_f2[f * 64 + lane]] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int f = wave + 4 * i;
            if (f < NF2) f2frag[f * 64 + lane] = tmp[i];
        }
        __syncthreads();
    }

    const int T = p.T, ntiles = p.ntiles;
    const float nm1 = (float)(T - 1), nT = (float)T;
    const float half_n0 = (float)ntiles * 0.5f;
    const int64_t rowstride = (int64_t)T * F;

#if BNN_PRIO_STAGGER
    // Waves that share a SIMD run the same program and fall into lockstep (both in their VALU phase, then both
    // wanting the matrix pipe).  Give odd hardware wave slots priority so the partner fills the gaps instead.
    if (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) __builtin_amdgcn_s_setprio(BNN_PRIO_STAGGER);
#endif
    // ---- wave-batches of 16 systems
    for (int64_t wb0 = b0 + (int64_t)wave * 16; wb0 < b1; wb0 += 64) {
        float skeep[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) skeep[k] = 0.0f;

        for (int q = 0; q < 4; ++q) {
            if (wb0 + 4 * q >= b1) break;  // wave-uniform
            const int64_t sys = wb0 + 4 * q + (c >> 2);
            const bool valid = sys < b1;
            const int64_t sysc = valid ? sys : b1 - 1;
            const float* rowp = p.x + sysc * rowstride + (int64_t)(c & 3) * F;
            const float* epin = nullptr;
            if constexpr (NOISY) {
                if (p.eps_in) epin = p.eps_in + (r * p.B + sysc) * rowstride + (int64_t)(c & 3) * F;
            }

            f32x4 mean0 = {0, 0, 0, 0}, m20 = {0, 0, 0, 0};
            float mean1 = 0.0f, m21 = 0.0f;

            // ---- building blocks of one 16-row tile ----------------------------------------------------------
            auto make_cur = [&](const XRaw<NK1>& raw, const XRaw<NK1>& nraw, const int it_cur) {
                XTile<NK1> cur = xtile<NK1>(raw, g);
                if constexpr (NOISY) {
                    // masks then add_input_noise (:486-504): masked columns become pure noise
                    XTile<NK1> ncur;
                    if (p.eps_in) {
                        ncur = xtile<NK1>(nraw, g);
                    } else {
                        // this lane's 11 columns 11g..11g+10 sit in Philox quads q0..q0+3 of row t (q0 = 11g/4)
                        const int t = 4 * it_cur + (c & 3);
                        const int q0 = (11 * g) >> 2, off = 11 * g - 4 * q0;
                        float f16[16];
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            f32x4 n4 = philox_sys4(TAG_IN, p.row_id0 + r, p.sys_id0 + sysc, t * 11 + q0 + b, p.seed);
                            f16[4 * b] = n4.x; f16[4 * b + 1] = n4.y; f16[4 * b + 2] = n4.z; f16[4 * b + 3] = n4.w;
                        }
#pragma unroll
                        for (int s = 0; s < NK1; ++s) {
                            float v0 = f16[s], v1 = f16[s + 1], v2 = f16[s + 2], v3 = f16[s + 3];
                            ncur.v[s] = off == 0 ? v0 : off == 1 ? v1 : off == 2 ? v2 : v3;
                        }
                    }
#pragma unroll
                    for (int s = 0; s < NK1; ++s) {
                        int col = 11 * g + s;
                        if (col < F) {
                            float xv = ((p.zero_mask >> col) & 1ull) ? 0.0f : cur.v[s];
                            cur.v[s] = xv + ncur.v[s] * in_scale[s];
                        }
                    }
                }
                return cur;
            };
            struct H3 { f32x4 m[3]; };
            struct H2 { f32x4 m[2]; };
            auto layer1 = [&](const XTile<NK1>& cur) {  // feature_nn.0 (bias rides in a k slot)
                H3 h = {{{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}};
#pragma unroll
                for (int s = 0; s < NK1; ++s)
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) h.m[mt] = mfma(wf[s * 3 + mt], cur.v[s], h.m[mt]);
                return h;
            };
            auto relu3 = [&](H3 h) {
                h.m[0] = relu4(h.m[0]); h.m[1] = relu4(h.m[1]); h.m[2] = relu4<2>(h.m[2]);
                return h;
            };
            auto layer2 = [&](const H3& h) {  // feature_nn.2
                H3 o;
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) o.m[mt] = (f32x4){wf[IB2 + mt * 4], wf[IB2 + mt * 4 + 1], wf[IB2 + mt * 4 + 2], wf[IB2 + mt * 4 + 3]};
#pragma unroll
                for (int ks = 0; ks < NKH; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) o.m[mt] = mfma(wf[IW2 + ks * 3 + mt], h.m[ks >> 2][ks & 3], o.m[mt]);
                return o;
            };
            auto layer3 = [&](const H3& h2) {  // feature_nn.4
                H2 y;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) y.m[mt] = (f32x4){wf[IB3 + mt * 4], wf[IB3 + mt * 4 + 1], wf[IB3 + mt * 4 + 2], wf[IB3 + mt * 4 + 3]};
#pragma unroll
                for (int ks = 0; ks < NKH; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) y.m[mt] = mfma(wf[IW3 + ks * 2 + mt], h2.m[ks >> 2][ks & 3], y.m[mt]);
                return y;
            };
            // torch.mean / torch.std over time (:418-419): Welford (fused updates) over this lane's timesteps, in tile order;
            // 1/(it+1) comes correctly rounded from a table
            auto pool = [&](const H2& y, const int it) {
                const float rcn = p.rcp_tab[it];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float dl = y.m[0][i] - mean0[i];
                    float mn = fmaf(dl, rcn, mean0[i]);
                    m20[i] = fmaf(dl, y.m[0][i] - mn, m20[i]);
                    mean0[i] = mn;
                }
                {
                    float dl = y.m[1][0] - mean1;
                    float mn = fmaf(dl, rcn, mean1);
                    m21 = fmaf(dl, y.m[1][0] - mn, m21);
                    mean1 = mn;
                }
            };
            // Loads stay where they are written: without the may-write barrier InstCombine folds phi(load, load) into a
            // load of phi(addresses) in front of the first use, and the machine scheduler sinks it further.
            auto prefetch = [&](XRaw<NK1>& raw, XRaw<NK1>& nraw, int it) {
                const int itc = it < ntiles ? it : ntiles - 1;  // past the end: re-read the last tile (no overrun)
                raw = load_x<NK1>(rowp + (int64_t)itc * 4 * F, g);
                if constexpr (NOISY) {
                    if (p.eps_in) nraw = load_x<NK1>(epin + (int64_t)itc * 4 * F, g);
                }
            };
            auto pin_loads = [&]() {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };

            XRaw<NK1> rawA, rawB, nrawA, nrawB;
            int it = 0;
            if constexpr (BNN_TWO_STREAMS && !NOISY) {
                // Two consecutive tiles travel through the layers together: while one stream's MFMAs occupy the
                // matrix pipe, the other stream's dependent ReLU / pool VALU work issues in their shadow, so the
                // pipe never waits on a layer boundary.  Pool order stays tile order (bit-identical results).
                prefetch(rawA, nrawA, 0);
                prefetch(rawB, nrawB, 1);
                pin_loads();
                for (; it + 1 < ntiles; it += 2) {
                    XTile<NK1> curA = make_cur(rawA, nrawA, it), curB = make_cur(rawB, nrawB, it + 1);
                    prefetch(rawA, nrawA, it + 2);   // one pair ahead, into the registers just consumed
                    prefetch(rawB, nrawB, it + 3);
                    pin_loads();
                    H3 hA = layer1(curA);
                    H3 hB = layer1(curB);
                    hA = relu3(hA);
                    H3 gA = layer2(hA);
                    hB = relu3(hB);
                    H3 gB = layer2(hB);
                    gA = relu3(gA);
                    H2 yA = layer3(gA);
                    gB = relu3(gB);
                    H2 yB = layer3(gB);
                    pool(yA, it);
                    pool(yB, it + 1);
                }
                if (it < ntiles) {  // odd tile count: rawA already holds the last tile
                    H2 y = layer3(relu3(layer2(relu3(layer1(make_cur(rawA, nrawA, it))))));
                    pool(y, it);
                }
            } else {
                // one tile at a time, next tile prefetched into a ping-pong pair of register sets
                auto do_tile = [&](const XRaw<NK1>& raw, const XRaw<NK1>& nraw, const int t) {
                    H2 y = layer3(relu3(layer2(relu3(layer1(make_cur(raw, nraw, t))))));
                    pool(y, t);
                };
                prefetch(rawA, nrawA, 0);
                pin_loads();
                for (; it + 1 < ntiles; it += 2) {
                    prefetch(rawB, nrawB, it + 1);
                    pin_loads();
                    do_tile(rawA, nrawA, it);
                    prefetch(rawA, nrawA, it + 2);
                    pin_loads();
                    do_tile(rawB, nrawB, it + 1);
                }
                if (it < ntiles) do_tile(rawA, nrawA, it);
            }

            // merge the 4 lanes of a quad (timesteps t = 4*it + (c&3)): equal-count Chan update, symmetric
            float mean[5] = {mean0[0], mean0[1], mean0[2], mean0[3], mean1};
            float m2[5] = {m20[0], m20[1], m20[2], m20[3], m21};
            float half_n = half_n0;
#pragma unroll
            for (int stage = 1; stage <= 2; stage <<= 1) {
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    float om = __shfl_xor(mean[k], stage), o2 = __shfl_xor(m2[k], stage);
                    float dl = om - mean[k];
                    float mm = (mean[k] + om) * 0.5f;
                    float qq = (m2[k] + o2) + (dl * dl) * half_n;
                    mean[k] = mm;
                    m2[k] = qq;
                }
                half_n = half_n * 2.0f;
            }

            // compute_summary_stats (:420-431) with the two randn_like draws
            f32x4 e1a, e2a;
            float e1b, e2b;
            if (p.eps) {
                const float* ep = p.eps + (r * p.B + sysc) * (2 * L);
                e1a = *reinterpret_cast<const f32x4*>(ep + 4 * g);
                e1b = ep[16 + g];
                e2a = *reinterpret_cast<const f32x4*>(ep + L + 4 * g);
                e2b = ep[L + 16 + g];
            } else {
                // The four lanes of a quad serve the same system and need the same four Philox blocks (quads g, 4, 5+g, 9
                // of that system's 40 normals): lane p of the quad generates block p, then the quad exchanges them
                // through the LDS crossbar (fp32 VALU time is matrix-pipe time on this chip; shuffles are not).
                const int64_t grow = p.row_id0 + r, gsys = p.sys_id0 + sysc;
                const int pq = c & 3;
                const int quad = pq == 0 ? g : pq == 1 ? 4 : pq == 2 ? 5 + g : 9;
                const f32x4 mine = philox_eps4(grow, gsys, quad, p.seed);
                const int qb = lane & ~3;
                const float pick = mine[g];  // lanes 1 and 3 of the quad only contribute component g
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e1a[k] = __shfl(mine[k], qb + 0);
                    e2a[k] = __shfl(mine[k], qb + 2);
                }
                e1b = __shfl(pick, qb + 1);
                e2b = __shfl(pick, qb + 3);
            }
            float snew[10];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                float e1 = k < 4 ? e1a[k] : e1b, e2 = k < 4 ? e2a[k] : e2b;
                float sample_mu = mean[k];
                float sd = sqrtf(m2[k] / nm1);   // torch.std (unbiased)
                float sample_var = sd * sd;      // **2
                float std_in_mu = sqrtf(sample_var / nT);
                float std_in_var = sqrtf((2.0f * (sample_var * sample_var)) / nm1);
                float mu_s = e1 * std_in_mu + sample_mu;
                float var_s = e2 * std_in_var + sample_var;
                snew[k] = mu_s;
                snew[5 + k] = sqrtf(fabsf(var_s) + 1e-5f);  // EPSILON (:337)
            }
            if (p.summary && valid && (c & 3) == 0) {
                float* sp = p.summary + (r * p.B + sys) * S2;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    int n = k < 4 ? 4 * g + k : 16 + g;
                    sp[n] = snew[k];
                    sp[L + n] = snew[5 + k];
                }
            }
            if ((c & 3) == q) {
#pragma unroll
                for (int k = 0; k < 10; ++k) skeep[k] = snew[k];
            }
        }

        // ---- regress_nn on 16 systems: column c <-> system wb0 + 4*(c&3) + (c>>2)
        const int64_t sysb = wb0 + 4 * (c & 3) + (c >> 2);
        const bool validb = sysb < b1;
        if constexpr (NOISY) {
            // add_summary_noise (:448-450)
            const int64_t sc = validb ? sysb : b1 - 1;
            if (p.eps_sum) {
                const float* es = p.eps_sum + (r * p.B + sc) * S2;
#pragma unroll
                for (int k = 0; k < 10; ++k) skeep[k] = skeep[k] + es[kmap_summary(k, g)] * sum_scale[k];
            } else {
                const int64_t grow = p.row_id0 + r, gsys = p.sys_id0 + sc;
#pragma unroll
                for (int kind = 0; kind < 2; ++kind) {
                    f32x4 a4n = philox_sys4(TAG_SUM, grow, gsys, kind * 5 + g, p.seed);
                    float bn = philox_sys4(TAG_SUM, grow, gsys, kind * 5 + 4, p.seed)[g];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) skeep[kind * 5 + rr] = skeep[kind * 5 + rr] + a4n[rr] * sum_scale[kind * 5 + rr];
                    skeep[kind * 5 + 4] = skeep[kind * 5 + 4] + bn * sum_scale[kind * 5 + 4];
                }
            }
        }
        const float* f2l = f2frag + lane;
        auto W2f = [&](int f) { return f2l[f * 64]; };
        f32x4 a4[3], a5[3], a6;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) a4[mt] = (f32x4){W2f(70 + mt * 4), W2f(71 + mt * 4), W2f(72 + mt * 4), W2f(73 + mt * 4)};
#pragma unroll
        for (int ks = 0; ks < 10; ++ks)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) a4[mt] = mfma(W2f(ks * 3 + mt), skeep[ks], a4[mt]);
        a4[0] = relu4(a4[0]); a4[1] = relu4(a4[1]); a4[2] = relu4<2>(a4[2]);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) a5[mt] = (f32x4){W2f(82 + mt * 4), W2f(83 + mt * 4), W2f(84 + mt * 4), W2f(85 + mt * 4)};
#pragma unroll
        for (int ks = 0; ks < NKH; ++ks)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) a5[mt] = mfma(W2f(30 + ks * 3 + mt), a4[ks >> 2][ks & 3], a5[mt]);
        a5[0] = relu4(a5[0]); a5[1] = relu4(a5[1]); a5[2] = relu4<2>(a5[2]);
        a6 = (f32x4){W2f(94), W2f(95), W2f(96), W2f(97)};
#pragma unroll
        for (int ks = 0; ks < NKH; ++ks) a6 = mfma(W2f(60 + ks), a5[ks >> 2][ks & 3], a6);

        if (g == 0 && validb) {
            // predict_instability + soft_clamp (:295-296, :437-442)
            float r0 = a6[0], r1 = a6[1];
            float mu = (0.5f * (tanhf(r0) + 1.0f)) * 8.0f + 4.0f;
            float sd = (0.5f * (tanhf(r1) + 1.0f)) * p.std_span + p.std_lo;
            if (bad_seed) mu = sd = __builtin_nanf("");
            const int64_t o = (r * p.B + sysb) * 2;
            *reinterpret_cast<f32x2*>(p.out + o) = (f32x2){mu, sd};
            if (p.pre_clamp) *reinterpret_cast<f32x2*>(p.pre_clamp + o) = (f32x2){r0, r1};
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Second feature_nn engine: v_mfma_f32_4x4x1_16b_f32 (bnn_layout.h, "second operand layout").
// lane = row, so there is no padding anywhere: 310 + 400 + 200 = 910 MFMAs of 8 cycles per 64 rows (113.75 pipe
// cycles per row against 148 for the 16x16x4 tiling).  Weights stream from an LDS image with broadcast
// ds_read_b128 (one read feeds four MFMAs; LDS reads do not occupy the fp32 pipe), activations never leave
// registers: a layer's accumulator registers are the next layer's B operands as they stand.
// A wave owns 16 systems at a time: lane l = system l>>2, timestep phase l&3; tile `it` = timesteps 4it..4it+3.
// Accumulation order per output = bias, then inputs in ascending order: the oracle's natural order.
// Built for the v50 column mask (31 live columns), quiet forward.  Everything after the time pool (sampled
// moments, regress_nn on the 16x16x4 path, soft_clamp) is shared with the first kernel.
// ------------------------------------------------------------------------------------------------
DEVINL f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

template <int CTRL>
DEVINL float quad_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

DEVINL void load_row31(const float* __restrict__ rp, float (&xv)[NLIVE4]) {
    xv[0] = rp[0];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4u*>(rp + 8 + 4 * q);
        xv[1 + 4 * q] = v.x; xv[2 + 4 * q] = v.y; xv[3 + 4 * q] = v.z; xv[4 + 4 * q] = v.w;
    }
    f32x2 t = *reinterpret_cast<const f32x2u*>(rp + 36);
    xv[29] = t.x; xv[30] = t.y;
}

constexpr int SCR4 = 2 * 16 * S2;  // floats of LDS scratch per wave: Philox normals + summaries of 16 systems

template <bool FUSED>
__global__ __launch_bounds__(256, 2) void bnn_multiswag4_kernel(const FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* flat = lds;                 // [FLAT_LDS] flat parameter vector + zero slot, later ...
    float* f2frag = lds;               // ... [NF2][64] regress_nn operands in fragment order
    float* zsh = lds + FLAT_LDS;       // [MAXK]
    float* wl = zsh + MAXK;            // [W4_PAD] feature_nn image for the 4x4x1 operands
    float* scr = wl + W4_PAD;          // [4][SCR4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;   // regress_nn (16x16x4) coordinates
    const int sl = lane >> 2, ph = lane & 3;  // feature_nn (4x4x1) coordinates: system in the wave-batch, timestep phase

    const int64_t id = blockIdx.x;
    const int e = (int)(id % p.J);
    const int64_t sub = id / p.J;
    const int ch = e % p.nch;
    const int64_t r = e / p.nch;
    const int64_t seg0 = (int64_t)ch * p.csz;
    const int64_t seg1 = (seg0 + p.csz < p.B) ? seg0 + p.csz : p.B;
    const int64_t b0 = seg0 + sub * p.spc;
    const int64_t b1 = (b0 + p.spc < seg1) ? b0 + p.spc : seg1;
    if (b0 >= b1) return;

#if BNN_STAMPS
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    bool bad_seed = false;
    if constexpr (FUSED) {
        int s = p.seed_idx[e];
        bad_seed = (s < 0 || s >= p.S);
        if (bad_seed) s = 0;
        const int K = p.K;
        if (tid < K) zsh[tid] = p.z2 ? p.z2[(int64_t)e * K + tid] : philox_z(TAG_Z2, p.draw_id0 + e, tid, p.seed);
        const float* wa = p.w_avg + (int64_t)s * D;
        const float* w2 = p.w2_avg + (int64_t)s * D;
        const float* pd = p.pre_D + (int64_t)s * D * K;
        __syncthreads();
        for (int i = tid; i < D; i += 256) {
            float z1v = p.z1 ? p.z1[(int64_t)e * D + i] : philox_z(TAG_Z1, p.draw_id0 + e, i, p.seed);
            flat[i] = draw_row_direct(wa, w2, pd, i, K, zsh, z1v, p.c1, p.c2, p.scale);
        }
    } else {
        const float* We = p.W + (int64_t)e * D;
        for (int i = tid; i < D; i += 256) flat[i] = We[i];
    }
    if (tid == 0) flat[ZERO_IDX] = 0.0f;
    __syncthreads();
    for (int i = tid; i < W4_PAD; i += 256) wl[i] = flat[p.tab_f4[i]];
    {
        constexpr int PER = (NF2 + 3) / 4;
        float tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int f = wave + 4 * i;
            tmp[i] = f < NF2 ? flat[p.tab_f2[f * 64 + lane]] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int f = wave + 4 * i;
            if (f < NF2) f2frag[f * 64 + lane] = tmp[i];
        }
        __syncthreads();
    }

    STAMP(0);  // prologue
    const int T = p.T, ntiles = p.ntiles;
    const float nm1 = (float)(T - 1), nT = (float)T;
    const float half_n0 = (float)ntiles * 0.5f;
    const int64_t rowstride = (int64_t)T * F;
    const f32x4* wqA1 = reinterpret_cast<const f32x4*>(wl + W4_L1A) + ph;
    const f32x4* wqB1 = reinterpret_cast<const f32x4*>(wl + W4_L1B) + ph;
    const f32x4* wqA2 = reinterpret_cast<const f32x4*>(wl + W4_L2A) + ph;
    const f32x4* wqB2 = reinterpret_cast<const f32x4*>(wl + W4_L2B) + ph;
    const f32x4* wqA3 = reinterpret_cast<const f32x4*>(wl + W4_L3A) + ph;
    const f32x4* wqB3 = reinterpret_cast<const f32x4*>(wl + W4_L3B) + ph;
    const f32x4* bq1 = reinterpret_cast<const f32x4*>(wl + W4_B1);
    const f32x4* bq2 = reinterpret_cast<const f32x4*>(wl + W4_B2);
    const f32x4* bq3 = reinterpret_cast<const f32x4*>(wl + W4_B3);
    float* epsscr = scr + wave * SCR4;
    float* sumscr = epsscr + 16 * S2;

    for (int64_t wb0 = b0 + (int64_t)wave * 16; wb0 < b1; wb0 += 64) {
        const int64_t sys = wb0 + sl;
        const bool valid = sys < b1;
        const int64_t sysc = valid ? sys : b1 - 1;
        const float* rowp = p.x + sysc * rowstride + (int64_t)ph * F;

        f32x4 mean[5], m2[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) { mean[n] = (f32x4){0, 0, 0, 0}; m2[n] = (f32x4){0, 0, 0, 0}; }

        float xv[NLIVE4];
        load_row31(rowp, xv);
        asm volatile("" ::: "memory");
        STAMP(1);  // batch setup + first row load issue
        for (int it = 0; it < ntiles; ++it) {
            // A operands are read one group of 20 MFMAs ahead of their use and the order is pinned with
            // sched_group_barrier (5 LDS reads, then 20 MFMAs): left alone, the scheduler issues each read one or two
            // MFMAs before its use and the LDS latency lands on the matrix pipe.
            // feature_nn.0 + ReLU: pairs of input columns (k0, k1): reads A(k0,m0) A(k0,m1) A(k1,m0) A(k1,m1) B(pair)
            f32x4 h[10];
            {
                constexpr int NP = (NLIVE4 + 1) / 2;  // 16 pairs, the last one holds only k = 30
                f32x4 q[NP][5];
#pragma unroll
                for (int n = 0; n < 10; ++n) h[n] = bq1[n];
                auto rd = [&](int kp) {
                    const int k0 = 2 * kp, k1 = 2 * kp + 1;
                    q[kp][0] = wqA1[(k0 * 2 + 0) * 4]; q[kp][1] = wqA1[(k0 * 2 + 1) * 4];
                    if (k1 < NLIVE4) { q[kp][2] = wqA1[(k1 * 2 + 0) * 4]; q[kp][3] = wqA1[(k1 * 2 + 1) * 4]; }
                    q[kp][4] = wqB1[kp * 4];
                };
                rd(0);
                __builtin_amdgcn_sched_group_barrier(0x100, 10 + 5, 0);
#pragma unroll
                for (int kp = 0; kp < NP; ++kp) {
                    if (kp + 1 < NP) rd(kp + 1);
                    const int k0 = 2 * kp, k1 = 2 * kp + 1;
                    const float b0 = xv[k0];
                    h[0] = mfma4(q[kp][0].x, b0, h[0]); h[1] = mfma4(q[kp][0].y, b0, h[1]); h[2] = mfma4(q[kp][0].z, b0, h[2]); h[3] = mfma4(q[kp][0].w, b0, h[3]);
                    h[4] = mfma4(q[kp][1].x, b0, h[4]); h[5] = mfma4(q[kp][1].y, b0, h[5]); h[6] = mfma4(q[kp][1].z, b0, h[6]); h[7] = mfma4(q[kp][1].w, b0, h[7]);
                    h[8] = mfma4(q[kp][4].x, b0, h[8]); h[9] = mfma4(q[kp][4].y, b0, h[9]);
                    if (k1 < NLIVE4) {
                        const float b1v = xv[k1];
                        h[0] = mfma4(q[kp][2].x, b1v, h[0]); h[1] = mfma4(q[kp][2].y, b1v, h[1]); h[2] = mfma4(q[kp][2].z, b1v, h[2]); h[3] = mfma4(q[kp][2].w, b1v, h[3]);
                        h[4] = mfma4(q[kp][3].x, b1v, h[4]); h[5] = mfma4(q[kp][3].y, b1v, h[5]); h[6] = mfma4(q[kp][3].z, b1v, h[6]); h[7] = mfma4(q[kp][3].w, b1v, h[7]);
                        h[8] = mfma4(q[kp][4].z, b1v, h[8]); h[9] = mfma4(q[kp][4].w, b1v, h[9]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
                }
            }
            STAMP(2);  // layer 1
#pragma unroll
            for (int n = 0; n < 10; ++n) h[n] = relu4(h[n]);
            // x of this tile is dead: fetch the next tile's rows into the same registers (one tile of work to land)
#if !(BNN_EXP & 2)  // timing experiment 2: no further x loads
            {
                const int itn = (it + 1 < ntiles) ? it + 1 : it;
                load_row31(rowp + (int64_t)itn * 4 * F, xv);
                asm volatile("" ::: "memory");
            }
#endif
            STAMP(3);  // relu 1 + load issue
            // feature_nn.2 + ReLU
            f32x4 h2[10];
            {
                constexpr int NP = H / 2;
                f32x4 q[NP][5];
#pragma unroll
                for (int n = 0; n < 10; ++n) h2[n] = bq2[n];
                auto rd = [&](int kp) {
                    const int k0 = 2 * kp, k1 = 2 * kp + 1;
                    q[kp][0] = wqA2[(k0 * 2 + 0) * 4]; q[kp][1] = wqA2[(k0 * 2 + 1) * 4];
                    q[kp][2] = wqA2[(k1 * 2 + 0) * 4]; q[kp][3] = wqA2[(k1 * 2 + 1) * 4];
                    q[kp][4] = wqB2[kp * 4];
                };
                rd(0);
                __builtin_amdgcn_sched_group_barrier(0x100, 10 + 5, 0);
#pragma unroll
                for (int kp = 0; kp < NP; ++kp) {
                    if (kp + 1 < NP) rd(kp + 1);
                    const int k0 = 2 * kp, k1 = 2 * kp + 1;
                    const float b0 = h[k0 >> 2][k0 & 3], b1v = h[k1 >> 2][k1 & 3];
                    h2[0] = mfma4(q[kp][0].x, b0, h2[0]); h2[1] = mfma4(q[kp][0].y, b0, h2[1]); h2[2] = mfma4(q[kp][0].z, b0, h2[2]); h2[3] = mfma4(q[kp][0].w, b0, h2[3]);
                    h2[4] = mfma4(q[kp][1].x, b0, h2[4]); h2[5] = mfma4(q[kp][1].y, b0, h2[5]); h2[6] = mfma4(q[kp][1].z, b0, h2[6]); h2[7] = mfma4(q[kp][1].w, b0, h2[7]);
                    h2[8] = mfma4(q[kp][4].x, b0, h2[8]); h2[9] = mfma4(q[kp][4].y, b0, h2[9]);
                    h2[0] = mfma4(q[kp][2].x, b1v, h2[0]); h2[1] = mfma4(q[kp][2].y, b1v, h2[1]); h2[2] = mfma4(q[kp][2].z, b1v, h2[2]); h2[3] = mfma4(q[kp][2].w, b1v, h2[3]);
                    h2[4] = mfma4(q[kp][3].x, b1v, h2[4]); h2[5] = mfma4(q[kp][3].y, b1v, h2[5]); h2[6] = mfma4(q[kp][3].z, b1v, h2[6]); h2[7] = mfma4(q[kp][3].w, b1v, h2[7]);
                    h2[8] = mfma4(q[kp][4].z, b1v, h2[8]); h2[9] = mfma4(q[kp][4].w, b1v, h2[9]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
                }
            }
            STAMP(4);  // layer 2
#pragma unroll
            for (int n = 0; n < 10; ++n) h2[n] = relu4(h2[n]);
            STAMP(5);  // relu 2
            // feature_nn.4: quads of inputs: reads A(k..k+3) + B(quad)
            f32x4 y[5];
            {
                constexpr int NQ = H / 4;
                f32x4 q[NQ][5];
#pragma unroll
                for (int n = 0; n < 5; ++n) y[n] = bq3[n];
                auto rd = [&](int kq) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) q[kq][cc] = wqA3[(4 * kq + cc) * 4];
                    q[kq][4] = wqB3[kq * 4];
                };
                rd(0);
                __builtin_amdgcn_sched_group_barrier(0x100, 5 + 5, 0);
#pragma unroll
                for (int kq = 0; kq < NQ; ++kq) {
                    if (kq + 1 < NQ) rd(kq + 1);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int k = 4 * kq + cc;
                        const float b = h2[k >> 2][k & 3];
                        y[0] = mfma4(q[kq][cc].x, b, y[0]); y[1] = mfma4(q[kq][cc].y, b, y[1]); y[2] = mfma4(q[kq][cc].z, b, y[2]); y[3] = mfma4(q[kq][cc].w, b, y[3]);
                        y[4] = mfma4(q[kq][4][cc], b, y[4]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
                }
            }
            STAMP(6);  // layer 3
            // torch.mean / torch.std over time (:418-419): Welford over this lane's timesteps
            const float rcn = p.rcp_tab[it];
#if BNN_EXP & 1  // timing experiment: pool replaced by integer ops (co-issue with the matrix pipe)
#pragma unroll
            for (int n = 0; n < 5; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    mean[n][i] = __builtin_bit_cast(float, __builtin_bit_cast(int, mean[n][i]) ^ __builtin_bit_cast(int, y[n][i]));
#else
#pragma unroll
            for (int n = 0; n < 5; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float dl = y[n][i] - mean[n][i];
                    float mn = fmaf(dl, rcn, mean[n][i]);
                    m2[n][i] = fmaf(dl, y[n][i] - mn, m2[n][i]);
                    mean[n][i] = mn;
                }
#endif
            STAMP(7);  // pool
        }

        // merge the 4 lanes of a quad: equal-count Chan update, symmetric (all four lanes end with the same bits)
        {
            float half_n = half_n0;
#pragma unroll
            for (int n = 0; n < 5; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float om = quad_perm<0xB1>(mean[n][i]), o2 = quad_perm<0xB1>(m2[n][i]);
                    float dl = om - mean[n][i];
                    float mm = (mean[n][i] + om) * 0.5f;
                    m2[n][i] = (m2[n][i] + o2) + (dl * dl) * half_n;
                    mean[n][i] = mm;
                }
            half_n = half_n * 2.0f;
#pragma unroll
            for (int n = 0; n < 5; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float om = quad_perm<0x4E>(mean[n][i]), o2 = quad_perm<0x4E>(m2[n][i]);
                    float dl = om - mean[n][i];
                    float mm = (mean[n][i] + om) * 0.5f;
                    m2[n][i] = (m2[n][i] + o2) + (dl * dl) * half_n;
                    mean[n][i] = mm;
                }
        }
        // The quad now holds four copies of the 20 pooled (mean, M2) pairs of its system: lane `ph` finishes
        // neurons 5ph..5ph+4 (compute_summary_stats :420-431), so the sqrt/divide sequences run once, not four times.
        float mymean[5], mym2[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float a0 = mean[j >> 2][j & 3], a1 = mean[(5 + j) >> 2][(5 + j) & 3], a2 = mean[(10 + j) >> 2][(10 + j) & 3],
                  a3 = mean[(15 + j) >> 2][(15 + j) & 3];
            float c0 = m2[j >> 2][j & 3], c1 = m2[(5 + j) >> 2][(5 + j) & 3], c2 = m2[(10 + j) >> 2][(10 + j) & 3],
                  c3 = m2[(15 + j) >> 2][(15 + j) & 3];
            mymean[j] = ph == 0 ? a0 : ph == 1 ? a1 : ph == 2 ? a2 : a3;
            mym2[j] = ph == 0 ? c0 : ph == 1 ? c1 : ph == 2 ? c2 : c3;
        }
        float e1[5], e2[5];
        if (p.eps) {
            const float* ep = p.eps + (r * p.B + sysc) * S2 + 5 * ph;
#pragma unroll
            for (int j = 0; j < 5; ++j) { e1[j] = ep[j]; e2[j] = ep[L + j]; }
        } else {
            // the system's 40 normals are ten Philox blocks: lane ph generates blocks ph, ph+4, ph+8 into LDS
            const int64_t grow = p.row_id0 + r, gsys = p.sys_id0 + sysc;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int qd = ph + 4 * t;
                if (qd < 10) *reinterpret_cast<f32x4*>(epsscr + sl * S2 + 4 * qd) = philox_eps4(grow, gsys, qd, p.seed);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 5; ++j) { e1[j] = epsscr[sl * S2 + 5 * ph + j]; e2[j] = epsscr[sl * S2 + L + 5 * ph + j]; }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float sample_mu = mymean[j];
            float sd = sqrtf(mym2[j] / nm1);   // torch.std (unbiased)
            float sample_var = sd * sd;        // **2
            float std_in_mu = sqrtf(sample_var / nT);
            float std_in_var = sqrtf((2.0f * (sample_var * sample_var)) / nm1);
            float mu_s = e1[j] * std_in_mu + sample_mu;
            float var_s = e2[j] * std_in_var + sample_var;
            float sd_s = sqrtf(fabsf(var_s) + 1e-5f);  // EPSILON (:337)
            sumscr[sl * S2 + 5 * ph + j] = mu_s;
            sumscr[sl * S2 + L + 5 * ph + j] = sd_s;
            if (p.summary && valid) {
                float* sp = p.summary + (r * p.B + sys) * S2 + 5 * ph + j;
                sp[0] = mu_s;
                sp[L] = sd_s;
            }
        }
        __builtin_amdgcn_wave_barrier();

        STAMP(8);  // merge + noise + finish
        // ---- regress_nn on the 16 systems of this wave-batch (16x16x4 path): column c <-> system wb0 + c
        const int64_t sysb = wb0 + c;
        const bool validb = sysb < b1;
        float skeep[10];
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) skeep[ks] = sumscr[c * S2 + kmap_summary(ks, g)];
        const float* f2l = f2frag + lane;
        auto W2f = [&](int f) { return f2l[f * 64]; };
        f32x4 a4[3], a5[3], a6;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) a4[mt] = (f32x4){W2f(70 + mt * 4), W2f(71 + mt * 4), W2f(72 + mt * 4), W2f(73 + mt * 4)};
#pragma unroll
        for (int ks = 0; ks < 10; ++ks)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) a4[mt] = mfma(W2f(ks * 3 + mt), skeep[ks], a4[mt]);
        a4[0] = relu4(a4[0]); a4[1] = relu4(a4[1]); a4[2] = relu4<2>(a4[2]);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) a5[mt] = (f32x4){W2f(82 + mt * 4), W2f(83 + mt * 4), W2f(84 + mt * 4), W2f(85 + mt * 4)};
#pragma unroll
        for (int ks = 0; ks < NKH; ++ks)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) a5[mt] = mfma(W2f(30 + ks * 3 + mt), a4[ks >> 2][ks & 3], a5[mt]);
        a5[0] = relu4(a5[0]); a5[1] = relu4(a5[1]); a5[2] = relu4<2>(a5[2]);
        a6 = (f32x4){W2f(94), W2f(95), W2f(96), W2f(97)};
#pragma unroll
        for (int ks = 0; ks < NKH; ++ks) a6 = mfma(W2f(60 + ks), a5[ks >> 2][ks & 3], a6);
        if (g == 0 && validb) {
            // predict_instability + soft_clamp (:295-296, :437-442)
            float r0 = a6[0], r1 = a6[1];
            float mu = (0.5f * (tanhf(r0) + 1.0f)) * 8.0f + 4.0f;
            float sd = (0.5f * (tanhf(r1) + 1.0f)) * p.std_span + p.std_lo;
            if (bad_seed) mu = sd = __builtin_nanf("");
            const int64_t o = (r * p.B + sysb) * 2;
            *reinterpret_cast<f32x2*>(p.out + o) = (f32x2){mu, sd};
#if !BNN_STAMPS
            if (p.pre_clamp) *reinterpret_cast<f32x2*>(p.pre_clamp + o) = (f32x2){r0, r1};
#endif
        }
        __builtin_amdgcn_wave_barrier();  // scratch is reused by the next wave-batch
        STAMP(9);  // regress_nn + store
    }
#if BNN_STAMPS
    if (p.pre_clamp && tid == 0) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.pre_clamp) + (int64_t)blockIdx.x * 12;
        for (int i = 0; i < 12; ++i) dst[i] = st_acc[i];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
__global__ void bnn_moments_kernel(const float* __restrict__ samples, int64_t R, int64_t B, double* __restrict__ mom, int accumulate) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (accumulate) { s0 = mom[b * 4]; s1 = mom[b * 4 + 1]; s2 = mom[b * 4 + 2]; s3 = mom[b * 4 + 3]; }
    for (int64_t r = 0; r < R; ++r) {
        f32x2 v = *reinterpret_cast<const f32x2*>(samples + (r * B + b) * 2);
        double mu = v.x, sd = v.y;
        s0 += mu; s1 += mu * mu; s2 += sd; s3 += sd * sd;
    }
    mom[b * 4] = s0; mom[b * 4 + 1] = s1; mom[b * 4 + 2] = s2; mom[b * 4 + 3] = s3;
}

// data_setup_kernel + StandardScaler.transform + .float(): one thread per (row, raw column j of the 32)
__global__ void bnn_feature_pack_kernel(const double* __restrict__ ts, const double* __restrict__ mass, const double* __restrict__ Xin,
                                        int64_t N, int T, const double* __restrict__ mean, const double* __restrict__ scale,
                                        double* __restrict__ X64, float* __restrict__ x32) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = N * T;
    if (Xin) {  // already packed: standardise only, one thread per element
        if (i >= rows * F) return;
        const int col = (int)(i % F);
        const double v = Xin[i];
        if (X64) X64[i] = v;
        if (x32) x32[i] = (float)((v - mean[col]) / scale[col]);
        return;
    }
    if (i >= rows * 32) return;
    const int j = (int)(i % 32);
    const int64_t row = i / 32, n = row / T;
    auto raw = [&](int c) -> double { return c < 26 ? ts[row * 26 + c] : mass[n * 3 + (c - 26)]; };
    double v;
    if (j < 29) v = raw(j);
    else v = (double)!isfinite(raw(j == 29 ? 3 : j == 30 ? 6 : 7));   // isnotfinite flags (regression.py:191-193)
    if (!isfinite(v)) v = 0.0;                                         // nan_to_num(posinf=0, neginf=0) (:195)
    // output column of raw column j: angles before it each add one column
    const bool angle = (j >= 11 && j <= 13) || (j >= 17 && j <= 19) || (j >= 23 && j <= 25);
    const int nbefore = (j > 11 ? (j < 14 ? j - 11 : 3) : 0) + (j > 17 ? (j < 20 ? j - 17 : 3) : 0) + (j > 23 ? (j < 26 ? j - 23 : 3) : 0);
    const int o = j + nbefore;
    auto put = [&](int col, double val) {
        if (X64) X64[row * F + col] = val;
        if (x32) x32[row * F + col] = (float)((val - mean[col]) / scale[col]);
    };
    if (angle) { put(o, cos(v)); put(o + 1, sin(v)); }
    else put(o, v);
}

__global__ void bnn_philox_fill_kernel(int kind, uint64_t seed, int64_t id0, int64_t n_rows, int64_t B, int64_t sys0, int width,
                                       float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (kind == 0 || kind == 1) {
        int64_t total = n_rows * width;
        if (i >= total) return;
        int64_t row = i / width;
        int el = (int)(i % width);
        out[i] = philox_z(kind == 0 ? TAG_Z1 : TAG_Z2, id0 + row, el, seed);
    } else if (kind == 2 || kind == 4) {
        int64_t total = n_rows * B * S2;
        if (i >= total) return;
        int el = (int)(i % S2);
        int64_t sys = (i / S2) % B, row = i / (S2 * B);
        out[i] = philox_sys4(kind == 2 ? TAG_EPS : TAG_SUM, id0 + row, sys0 + sys, el >> 2, seed)[el & 3];
    } else {  // kind 3: eps_in [n_rows, B, T = width, 41]
        const int T = width;
        int64_t per = (int64_t)T * F, total = n_rows * B * per;
        if (i >= total) return;
        int col = (int)(i % F), t = (int)((i / F) % T);
        int64_t sys = (i / per) % B, row = i / (per * B);
        out[i] = philox_sys4(TAG_IN, id0 + row, sys0 + sys, t * 11 + (col >> 2), seed)[col & 3];
    }
}

__global__ void bnn_philox_raw_kernel(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int64_t n,
                                      uint32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 r = philox4x32_10(make_uint4(c0 + (uint32_t)i, c1, c2, c3), make_uint2(k0, k1));
    out[i * 4] = r.x; out[i * 4 + 1] = r.y; out[i * 4 + 2] = r.z; out[i * 4 + 3] = r.w;
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
    Tables tab[2];  // [0] = arch mask, [1] = all 41 columns (noisy forward)
    int16_t* d_f1[2] = {nullptr, nullptr};
    int16_t* d_f2[2] = {nullptr, nullptr};
    int16_t* d_f4 = nullptr;  // 4x4x1 image table (v50 mask only)
    bool use_v4 = false;      // feature_nn on the 4x4x1 engine (v50 mask, quiet forward)
    float* d_rcp = nullptr;  // [RCP_N] 1/(i+1)
    int device = 0;
};

constexpr int RCP_N = 4096;  // supports T up to 16384 timesteps

static int check_arch(const bnn_arch* a) {
    if (!a) return fail(BNN_ERR_INVALID, "arch is NULL");
    if (a->n_features != F || a->hidden != H || a->latent != L)
        return fail(BNN_ERR_UNSUPPORTED, "only the 41->40->40->20 / 40->40->40->2 network of the pretrained ensemble is built");
    if (a->zero_mask >> F) return fail(BNN_ERR_INVALID, "zero_mask has bits beyond column 40");
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

int bnn_param_count(const bnn_arch* arch) {
    int rc = check_arch(arch);
    return rc ? rc : D;
}

int bnn_plan_create(const bnn_arch* arch, bnn_plan** out) {
    int rc = check_arch(arch);
    if (rc) return rc;
    if (!out) return fail(BNN_ERR_INVALID, "out is NULL");
    bnn_plan* pl = new bnn_plan();
    pl->arch = *arch;
    pl->tab[0] = build_tables(arch->zero_mask, false);
    pl->tab[1] = build_tables(arch->zero_mask, true);
    if (hipGetDevice(&pl->device) != hipSuccess) {
        delete pl;
        return fail(BNN_ERR_NO_DEVICE, "no HIP device");
    }
    for (int v = 0; v < 2; ++v) {
        size_t n1 = pl->tab[v].f1.size() * sizeof(int16_t), n2 = pl->tab[v].f2.size() * sizeof(int16_t);
        if (hipMalloc(&pl->d_f1[v], n1) != hipSuccess || hipMalloc(&pl->d_f2[v], n2) != hipSuccess ||
            hipMemcpy(pl->d_f1[v], pl->tab[v].f1.data(), n1, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(pl->d_f2[v], pl->tab[v].f2.data(), n2, hipMemcpyHostToDevice) != hipSuccess) {
            bnn_plan_destroy(pl);
            return fail(BNN_ERR_HIP, "plan table upload failed");
        }
    }
    if (!pl->tab[0].f4.empty()) {
        size_t n4 = pl->tab[0].f4.size() * sizeof(int16_t);
        if (hipMalloc(&pl->d_f4, n4) != hipSuccess || hipMemcpy(pl->d_f4, pl->tab[0].f4.data(), n4, hipMemcpyHostToDevice) != hipSuccess) {
            bnn_plan_destroy(pl);
            return fail(BNN_ERR_HIP, "plan table upload failed");
        }
        const char* kv = getenv("BNN_CHAOS_KERNEL");  // "16x16" forces the first engine (A/B runs)
        pl->use_v4 = !(kv && std::string(kv) == "16x16");
    }
    {
        std::vector<float> rc(RCP_N);
        for (int i = 0; i < RCP_N; ++i) rc[i] = 1.0f / (float)(i + 1);
        if (hipMalloc(&pl->d_rcp, RCP_N * sizeof(float)) != hipSuccess ||
            hipMemcpy(pl->d_rcp, rc.data(), RCP_N * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            bnn_plan_destroy(pl);
            return fail(BNN_ERR_HIP, "plan table upload failed");
        }
    }
    *out = pl;
    return 0;
}

int bnn_plan_destroy(bnn_plan* pl) {
    if (!pl) return 0;
    for (int v = 0; v < 2; ++v) {
        if (pl->d_f1[v]) (void)hipFree(pl->d_f1[v]);
        if (pl->d_f2[v]) (void)hipFree(pl->d_f2[v]);
    }
    if (pl->d_rcp) (void)hipFree(pl->d_rcp);
    if (pl->d_f4) (void)hipFree(pl->d_f4);
    delete pl;
    return 0;
}

int bnn_plan_layer_order(const bnn_plan* pl, int layer, int noisy, int32_t* host_order, int cap) {
    if (!pl || layer < 0 || layer > 5) return fail(BNN_ERR_INVALID, "bad plan/layer");
    std::vector<int32_t> o = pl->tab[noisy ? 1 : 0].order[layer];
    if (!noisy && pl->use_v4 && layer < 3) {  // 4x4x1 engine: bias first, then inputs in ascending order
        o.clear();
        for (int k = 0; k < (layer == 0 ? F : H); ++k)
            if (layer != 0 || !((pl->arch.zero_mask >> k) & 1ull)) o.push_back(k);
    }
    if (host_order)
        for (int i = 0; i < (int)o.size() && i < cap; ++i) host_order[i] = o[i];
    return (int)o.size();
}

int bnn_layer_order(const bnn_arch* arch, int layer, int noisy, int32_t* host_order, int cap) {
    int rc = check_arch(arch);
    if (rc) return rc;
    if (layer < 0 || layer > 5) return fail(BNN_ERR_INVALID, "layer must be 0..5");
    Tables t = build_tables(arch->zero_mask, noisy != 0);
    const std::vector<int32_t>& o = t.order[layer];
    if (host_order)
        for (int i = 0; i < (int)o.size() && i < cap; ++i) host_order[i] = o[i];
    return (int)o.size();
}

int bnn_fragment_table(const bnn_arch* arch, int noisy, int which, int16_t* host_table, int cap) {
    int rc = check_arch(arch);
    if (rc) return rc;
    if (which != 1 && which != 2) return fail(BNN_ERR_INVALID, "which must be 1 or 2");
    Tables t = build_tables(arch->zero_mask, noisy != 0);
    const std::vector<int16_t>& v = which == 1 ? t.f1 : t.f2;
    if (host_table)
        for (int i = 0; i < (int)v.size() && i < cap; ++i) host_table[i] = v[i];
    return (int)v.size();
}

static int pick_spc(const bnn_grid* g, int64_t csz) {
    if (g->systems_per_block > 0) return g->systems_per_block;
    // The per-workgroup prologue (flat vector -> LDS operand images) is amortised over the block: prefer big blocks
    // (512 systems: +1.7 % over 256 at configs[1]) as long as the grid still fills 256 CUs x 2 several times over.
    for (int spc : {512, 256, 128}) {
        int64_t nsub = (csz + spc - 1) / spc;
        if (nsub * (int64_t)g->J >= 4096) return spc;
    }
    return 64;
}

static int launch_forward(const bnn_plan* pl, const bnn_grid* g, FwdParams& p, bool fused, bool noisy, void* stream) {
    if (!pl || !g) return fail(BNN_ERR_INVALID, "plan/grid is NULL");
    if (g->B < 0 || g->J < 0 || g->nchunks < 1) return fail(BNN_ERR_INVALID, "negative size");
    if (g->J % g->nchunks) return fail(BNN_ERR_INVALID, "J must be a multiple of nchunks");
    if (g->T < 8 || (g->T % 4) || g->T / 4 > RCP_N) return fail(BNN_ERR_UNSUPPORTED, "T must be a multiple of 4 in [8, 16384]");
    if (g->systems_per_block < 0 || (g->systems_per_block % 64)) return fail(BNN_ERR_INVALID, "systems_per_block must be a multiple of 64");
    if (g->B == 0 || g->J == 0) return 0;
    if (!p.x || !p.out) return fail(BNN_ERR_INVALID, "x/out is NULL");
    if (p.draw_id0 % g->nchunks) return fail(BNN_ERR_INVALID, "draw_id0 must be a multiple of nchunks");
    const int v = noisy ? 1 : 0;
    p.B = g->B; p.T = g->T; p.ntiles = g->T / 4; p.J = g->J; p.nch = g->nchunks;
    p.csz = (g->B + g->nchunks - 1) / g->nchunks;
    p.spc = pick_spc(g, p.csz);
    p.row_id0 = p.draw_id0 / g->nchunks;
    p.tab_f1 = pl->d_f1[v]; p.tab_f2 = pl->d_f2[v]; p.tab_f4 = pl->d_f4; p.rcp_tab = pl->d_rcp;
    p.zero_mask = pl->arch.zero_mask;
    p.std_lo = pl->arch.lowest_std; p.std_span = (float)(6.0 - (double)pl->arch.lowest_std);
    const int64_t nsub = (p.csz + p.spc - 1) / p.spc;
    const int64_t nblk = nsub * g->J;
    if (nblk > 0x7fffffffLL) return fail(BNN_ERR_RANGE, "grid too large; split the draws");
    const size_t shmem = sizeof(float) * (FLAT_LDS + MAXK);
    static_assert(NF2 * 64 <= FLAT_LDS, "regress_nn fragments overwrite the flat vector in place");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nblk), block(256);
    const int nk1 = pl->tab[v].nk1;
#define LAUNCH(NK, NZ, FU)                                                                                         \
    do {                                                                                                           \
        static std::once_flag once;                                                                                \
        std::call_once(once, [] {                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bnn_multiswag_kernel<NK, NZ, FU>),            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                     \
        });                                                                                                        \
        hipLaunchKernelGGL((bnn_multiswag_kernel<NK, NZ, FU>), grid, block, shmem, st, p);                         \
    } while (0)
    if (!noisy && nk1 == 8 && pl->use_v4) {
        const size_t shmem4 = sizeof(float) * (FLAT_LDS + MAXK + W4_PAD + 4 * SCR4);
#define LAUNCH4(FU)                                                                                                \
    do {                                                                                                           \
        static std::once_flag once;                                                                                \
        std::call_once(once, [] {                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bnn_multiswag4_kernel<FU>),                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                     \
        });                                                                                                        \
        hipLaunchKernelGGL((bnn_multiswag4_kernel<FU>), grid, block, shmem4, st, p);                               \
    } while (0)
        if (fused) LAUNCH4(true); else LAUNCH4(false);
#undef LAUNCH4
    } else if (noisy) {
        LAUNCH(11, true, false);
    } else if (nk1 == 8) {
        if (fused) LAUNCH(8, false, true); else LAUNCH(8, false, false);
    } else {
        if (fused) LAUNCH(11, false, true); else LAUNCH(11, false, false);
    }
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return 0;
}

static int draw_consts(int K, float scale, float* c1, float* c2) {
    if (K < 2 || K > MAXK) return fail(BNN_ERR_RANGE, "SWAG rank K must be in [2, 32]");
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
    dim3 grid((D + 255) / 256, J), block(256);
    hipLaunchKernelGGL(bnn_swag_draw_kernel, grid, block, 0, (hipStream_t)stream, w_avg, w2_avg, pre_D, S, K, seed_idx, z1, z2, c1,
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
    int rc = draw_consts(K, scale, &p.c1, &p.c2);
    if (rc) return rc;
    p.scale = scale; p.K = K; p.S = S;
    p.x = x; p.w_avg = w_avg; p.w2_avg = w2_avg; p.pre_D = pre_D; p.seed_idx = seed_idx; p.z1 = z1; p.z2 = z2; p.eps = eps;
    p.seed = philox_seed; p.draw_id0 = draw_id0; p.sys_id0 = system_id0;
    p.out = out; p.pre_clamp = pre_clamp; p.summary = summary;
    return launch_forward(plan, grid, p, true, false, stream);
}

int bnn_feature_pack_f64(const double* tseries, const double* mass, const double* X64_in, int64_t N, int32_t T, const double* mean,
                         const double* scale, double* X64_out, float* x32_out, void* stream) {
    if (N < 0 || T < 1) return fail(BNN_ERR_INVALID, "bad N/T");
    if (N == 0) return 0;
    if (!tseries && !X64_in) return fail(BNN_ERR_INVALID, "need tseries (+mass) or X64_in");
    if (tseries && !mass) return fail(BNN_ERR_INVALID, "tseries needs mass");
    if (!X64_out && !x32_out) return fail(BNN_ERR_INVALID, "no output requested");
    if (x32_out && (!mean || !scale)) return fail(BNN_ERR_INVALID, "x32_out needs mean and scale");
    const int64_t total = N * T * (tseries ? 32 : F);
    hipLaunchKernelGGL(bnn_feature_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tseries, mass,
                       tseries ? nullptr : X64_in, N, (int)T, mean, scale, X64_out, x32_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_moments_f64(const float* samples, int64_t R, int64_t B, double* moments, int32_t accumulate, void* stream) {
    if (!samples || !moments || R < 0 || B < 0) return fail(BNN_ERR_INVALID, "bad argument");
    if (B == 0) return 0;
    hipLaunchKernelGGL(bnn_moments_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, samples, R, B, moments,
                       accumulate);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bnn_philox_normal_f32(int32_t kind, uint64_t philox_seed, int64_t id0, int64_t n_rows, int64_t B, int64_t system_id0, int32_t width,
                          float* out, void* stream) {
    if (!out || kind < 0 || kind > 4 || n_rows < 0) return fail(BNN_ERR_INVALID, "bad argument");
    int64_t total = (kind == 2 || kind == 4) ? n_rows * B * S2 : kind == 3 ? n_rows * B * (int64_t)width * F : n_rows * (int64_t)width;
    if (total == 0) return 0;
    hipLaunchKernelGGL(bnn_philox_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kind, philox_seed,
                       id0, n_rows, B, system_id0, width, out);
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

}  // extern "C"
