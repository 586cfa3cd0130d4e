// bnn_forward.hip.h -- the forward kernel of the MultiSWAG path: feature_nn on
// v_mfma_f32_4x4x1_16b_f32 with register-resident weights, time pool, sampled moments, regress_nn, soft_clamp, optionally the
// in-prologue SWAG draw (FUSED), forward(noisy_val=True) (NOISY, XNOISE), the fused statistics tail (STATS), fix_megno (MEGNO).
// Included by the bnn_fwd_*.hip translation units, each of which instantiates a few of the template's forms.
//
// lane = row, so there is no padding anywhere: 310 + 400 + 200 = 910 MFMAs of 8 cycles per 64 rows (113.75 pipe cycles per
// row against 148 for a 16x16x4 tiling).  The weights are REGISTER-RESIDENT: with CBSZ = 4 the MFMA broadcasts the A operand of
// block ABID to all 16 blocks, so one VGPR carries the A operands of 16 different MFMAs (lanes 4a..4a+3 = the four neurons of
// MFMA 16R + a) and the whole of feature_nn is 58 VGPRs (WR<KIN>), loaded once per workgroup; the tile loop reads no weights
// from anywhere (round 2 streamed them from LDS images: 253 ds_read_b128 + 212 s_waitcnt per 910-MFMA tile, LDS 45 % busy).
// Activations never leave registers: a layer's accumulator registers are the next layer's B operands as they stand.
// A wave owns 16 systems at a time: lane l = system l>>2, timestep phase l&3; tile `it` = timesteps 4it..4it+3.
// Accumulation order per output = bias, then inputs in ascending order: the oracle's natural order.
// KIN = 31: the v50 column mask (31 live columns); KIN = 41: any mask (whole rows, zero weights on masked columns).
// The 4x4x1 MFMA holds the SIMD's vector issue port for all of its 8 cycles (profiles/r02_coexec2_probe.txt): nothing
// co-issues with it, from either wave of the SIMD, so the kernel's time is the SUM of its matrix and vector instructions and the
// loop below is written to need as few vector instructions as the arithmetic allows.
#pragma once
#include "bnn_common.hip.h"
#include "bnn_stats.hip.h"

// Build-time switches of the tile loop (A/B'd on one box, scripts/ab_variants.py; both on: -4.6 % at configs[2]):
// BNN_RELU_BATCH: the 40 ReLUs of a layer in one run behind a scheduling barrier; BNN_BIAS_PREFETCH: accumulators are initialised
// with their biases a layer ahead (31-column forms only: the 41-column forms have no registers to spare).
// BNN_ABLATE (profiling builds ONLY, results are wrong by construction; scripts/ablate_r03.sh): bit 0 drops the ReLUs, bit 1 the
// Welford pool, bit 2 the per-tile x loads (tile 0's rows are reused), bit 3 everything after the tile loop (merge, sampled
// moments, regress_nn, soft_clamp).  The time each removal saves is that part's cost in the real kernel (profiles/r03_ablation_c3.txt).
#ifndef BNN_ABLATE
#define BNN_ABLATE 0
#endif

#ifndef BNN_BIAS_PREFETCH
#define BNN_BIAS_PREFETCH 1
#endif
#ifndef BNN_RELU_BATCH
#define BNN_RELU_BATCH 1
#endif

namespace bnn {

template <int KIN>
DEVINL void load_row(const float* __restrict__ rp, float (&xv)[KIN]) {
    if constexpr (KIN == 31) {  // v50 mask: columns 0, 8..37
        xv[0] = rp[0];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4u*>(rp + 8 + 4 * q);
            xv[1 + 4 * q] = v.x; xv[2 + 4 * q] = v.y; xv[3 + 4 * q] = v.z; xv[4 + 4 * q] = v.w;
        }
        f32x2 t = *reinterpret_cast<const f32x2u*>(rp + 36);
        xv[29] = t.x; xv[30] = t.y;
    } else {  // any mask: the whole row
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4u*>(rp + 4 * q);
            xv[4 * q] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
        }
        xv[40] = rp[40];
    }
}

constexpr int MEGSCR = 32;         // LDS floats per wave: [16 systems][megno mean, megno std] (fix_megno forms)
constexpr int NSC4 = 96 + 2 * 56;  // LDS floats of the noisy forward: exp(logvar/2) for 41 inputs + 40 summaries (padded to 96), then
                                   // per 6-column noise block, padded to 8: the input scales [7][8] and the column keep-masks [7][8]

constexpr int YBUF4 = 2 * 4 * 5 * 64;   // TSPLIT: float4 slots of the latent exchange, [buffer][wave][latent group][lane]

template <int KIN, bool TSPLIT = false>
constexpr size_t fwd_lds_bytes() { return sizeof(float) * (FLAT_LDS + MAXK + BIAS_PAD + 4 * SCR4 + NSC4 + 4 * MEGSCR + (TSPLIT ? 4 * YBUF4 : 0)); }

// MEGNO: hparams['fix_megno'] (spock_reg_model.py:360-362, 480-491, 509-510): the summary gains the time mean and unbiased std of
// the RAW MEGNO column (read before the masks and before any noise), pooled with the same per-lane Welford + quad merge as the
// latents; regress_nn.0 takes 42 inputs (an 11th k-step), the flat vector is Lay<true> (d = 7665).
// XNOISE (noisy forms): the input and summary noise come from explicit tensors (parity mode: the reference's numbers) instead of
// in-kernel Philox; a template parameter rather than a wave-uniform branch per noise block (7 branches + dead loads per tile).
// TSPLIT (small grids: the evaluation scripts' per-chunk calls, 15 .. 3 000 rows under ONE draw -- figures/multiswag_5_planet.py:295-298,
// figures/main_figures.py:154-156): a workgroup owns 16 systems and its four waves split the TILES between them (wave w takes tiles
// w, w + 4, ...), so a call whose whole grid is a few hundred wave-batches spreads over four times as many SIMDs and a batch's 25 tiles
// take 7 rounds instead of 25.  The pool must not change: each round's latents go through LDS to wave 0, which runs the SAME per-lane
// Welford chain over the tiles in ascending order -- bit-identical outputs to the plain form (tested), which is why this is a launch
// form and not another engine.  One barrier per round (double-buffered exchange).
template <int KIN, bool FUSED, bool NOISY, bool STATS, bool MEGNO = false, bool XNOISE = false, bool TSPLIT = false>
__global__ __launch_bounds__(256, 2) void bnn_forward_kernel(const FwdParams p) {
    using WRL = WR<KIN>;
    using Y = Lay<MEGNO>;
    constexpr int D = Y::D;                                  // shadows the fix_megno = False constant
    static_assert(!MEGNO || !STATS, "the fused statistics tail is not built for fix_megno");
    static_assert(!XNOISE || NOISY, "explicit input / summary noise belongs to the noisy forms");
    constexpr bool PREF = BNN_BIAS_PREFETCH && KIN == 31 && !MEGNO;   // the other forms have no registers to spare for it
    constexpr bool RBATCH = BNN_RELU_BATCH != 0;
    static_assert(!NOISY || (KIN == F && !FUSED && !STATS), "the noisy forward multiplies all 41 columns and takes materialised weights");
    static_assert(!TSPLIT || (!NOISY && !MEGNO && !STATS), "the tile-split launch form is built for the quiet forward only");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* flat = lds;                 // [FLAT_LDS] flat parameter vector + zero slot, later ...
    float* f2frag = lds;               // ... [NF2][64] regress_nn operands in fragment order
    float* zsh = lds + FLAT_LDS;       // [MAXK]
    float* wl = zsh + MAXK;            // [BIAS_PAD] feature_nn biases [b1 | b2 | b3], 16-byte aligned rows of 4
    float* scr = wl + BIAS_PAD;        // [4][SCR4]
    float* nsc = scr + 4 * SCR4;       // [NSC4] (NOISY only)
    float* megscr = nsc + NSC4;        // [4][MEGSCR] (MEGNO only)
    f32x4* ybuf = reinterpret_cast<f32x4*>(megscr + 4 * MEGSCR);   // [2][4][5][64] (TSPLIT only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl0 = lane >> 2, ph0 = lane & 3;  // feature_nn (4x4x1) coordinates: system in the wave-batch, timestep phase

    // work item: draw e, block `sub` of its chunk of systems (torch.chunk semantics); XCD-aware draw-fastest order
    // (work_item, bnn_common.hip.h): the workgroups resident on one XCD work on the SAME systems under different draws, so x
    // comes from HBM once and from that XCD's L2 after that.
    const WorkItem wi = work_item(p);
    const int e = wi.e;
    const int64_t sub = wi.sub;
    const int ch = e % p.nch;
    const int64_t r = e / p.nch;  // output row
    // the draw's chunk of systems (torch.chunk over the cB systems of the WHOLE batch, of which this call holds rows [coff, coff + B))
    const int64_t g0 = (int64_t)ch * p.csz - p.coff, gend = p.cB - p.coff;
    const int64_t seg0 = g0 > 0 ? g0 : 0;
    int64_t seg1 = (g0 + p.csz < gend) ? g0 + p.csz : gend;
    seg1 = seg1 < p.B ? seg1 : p.B;
    const int64_t b0 = seg0 + sub * p.spc;
    const int64_t b1 = (b0 + p.spc < seg1) ? b0 + p.spc : seg1;
    if (b0 >= b1) return;

    // ---- prologue: flat parameter vector of draw e -> LDS -> weight registers, bias image, regress_nn fragments
    // The gather tables (wave-invariant: 58 + 7 small loads per lane) are requested FIRST, so that they travel while the parameter
    // vector is copied / sampled -- one memory round trip instead of three in a row (a small grid's whole run time is a few of them).
    constexpr int NF2 = Y::NF2;
    constexpr int PER = (NF2 + 3) / 4;
    // (Small grids only, and not in the FUSED forms, whose prologue is the draw.)
    int twr[WRL::NR], idx[PER];
    auto fetch_tables = [&]() {
#pragma unroll
        for (int R = 0; R < WRL::NR; ++R) twr[R] = p.tab_wr[R * 64 + lane];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int f = wave + 4 * i;
            idx[i] = p.tab_f2[(f < NF2 ? f : NF2 - 1) * 64 + lane];
        }
    };
    constexpr bool TABLES_FIRST = !FUSED && TSPLIT;   // (in the plain form the early request measured 1.3 % SLOWER at configs[2], same box)
    if constexpr (TABLES_FIRST) {
        fetch_tables();
        asm volatile("" ::: "memory");
    }
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
        // (Measured, round 6: this loop is bound by its ~140 vector instructions per element at one wave per SIMD -- 12 us of a 15-row call's
        // 50 us kernel with explicit normals, 19 us with in-kernel Philox -- not by its memory round trips: five elements' rows requested
        // ahead of the arithmetic changed nothing, and neither did 16-byte row loads.)
        for (int i = tid; i < D; i += 256) {
            float z1v = p.z1 ? p.z1[(int64_t)e * D + i] : philox_z(TAG_Z1, p.draw_id0 + e, i, p.seed);
            flat[i] = draw_row_direct(wa, w2, pd, i, K, zsh, z1v, p.c1, p.c2, p.scale);
        }
    } else {
        const float* We = p.W + (int64_t)e * D;
        for (int i = tid; i < D; i += 256) flat[i] = We[i];
    }
    if (tid == 0) flat[Y::ZERO] = 0.0f;
    if constexpr (!TABLES_FIRST) fetch_tables();
    __syncthreads();
    // feature_nn weights -> registers (every wave holds the same 58): register R, lane 4a+i = W[neuron 4n+i][input k] of the
    // layer's MFMA number m = 16R + a = k * groups + n (bnn_layout.h, WR<KIN>); biases -> a small LDS image
    float wr[WRL::NR];
#pragma unroll
    for (int R = 0; R < WRL::NR; ++R) wr[R] = flat[twr[R]];
    if (tid < 2 * H + L) wl[tid] = flat[tid < H ? Y::B1 + tid : tid < 2 * H ? Y::B2 + (tid - H) : Y::B3 + (tid - 2 * H)];
    if constexpr (NOISY) {  // exp(input_noise_logvar/2) (:445), exp(summary_noise_logvar/2) (:449)
        if (tid < F + Y::SM) nsc[tid] = expf(flat[Y::INLV + tid] / 2.0f);
        if (tid < 56) {     // the same input scales per noise block, and 1.0 / 0.0 keep-factors for kept / zeroed columns
            constexpr int NPB = (BNN_NIN16 && !XNOISE) ? NIN16_PER_BLOCK : NIN_PER_BLOCK;
            const int col = NPB * (tid >> 3) + (tid & 7);
            const bool live = (tid & 7) < NPB && col < F;
            nsc[96 + tid] = live ? expf(flat[Y::INLV + col] / 2.0f) : 0.0f;
            nsc[96 + 56 + tid] = (live && !((p.zero_mask >> col) & 1ull)) ? 1.0f : 0.0f;
        }
    }
    {   // regress_nn operands replace the flat vector in place: gather to registers, barrier, write (table entries: fetched at the top,
        // branch-free, clamped).
        float tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) tmp[i] = flat[idx[i]];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int f = wave + 4 * i;
            if (f < NF2) f2frag[f * 64 + lane] = tmp[i];
        }
        __syncthreads();
    }

    const int T = p.T, ntiles = p.ntiles;
    const float nm1 = (float)(T - 1), nT = (float)T;
    const float half_n0 = (float)ntiles * 0.5f;
    const int64_t rowstride = (int64_t)T * F;
    const f32x4* bq1 = reinterpret_cast<const f32x4*>(wl);
    const f32x4* bq2 = reinterpret_cast<const f32x4*>(wl + H);
    const f32x4* bq3 = reinterpret_cast<const f32x4*>(wl + 2 * H);
    float* epsscr = scr + wave * SCR4;
    float* sumscr = epsscr + 16 * S2;

    for (int64_t wb0 = b0 + (TSPLIT ? 0 : (int64_t)wave * 16); wb0 < b1; wb0 += (TSPLIT ? 16 : 64)) {
        if constexpr (TSPLIT) __syncthreads();   // (wave 0 may still be in the previous batch's tail, reading the exchange buffer)
        const int64_t sys0 = wb0 + sl0;
        const bool valid0 = sys0 < b1;
        const int64_t sysc0 = valid0 ? sys0 : b1 - 1;
        const float* rowp = p.x + sysc0 * rowstride + (int64_t)ph0 * F;

        f32x4 mean[5], m2[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) { mean[n] = (f32x4){0, 0, 0, 0}; m2[n] = (f32x4){0, 0, 0, 0}; }

        float xv[KIN];
        load_row<KIN>(rowp + (TSPLIT ? (int64_t)(wave < ntiles ? wave : 0) * 4 * F : 0), xv);   // this wave's first tile
        float xmeg = 0.0f, gmean = 0.0f, gm2 = 0.0f;   // MEGNO: the raw column 7 of this lane's row, its running mean and M2
        if constexpr (MEGNO) xmeg = rowp[MEGNO_COL];
        asm volatile("" ::: "memory");
        // Accumulators of the three layers.  An accumulator chain starts at its bias (C operand of its first MFMA), read from the LDS
        // bias image.  With PREF the reads run a layer ahead of their first use: feature_nn.2's biases are fetched at the top of
        // feature_nn.0, feature_nn.4's behind feature_nn.2's last MFMA (they land while its ReLUs issue), the NEXT tile's
        // feature_nn.0 biases at the top of feature_nn.4 -- no MFMA ever waits on an LDS read.  sched_barrier(0) pins each group of
        // reads (and each batch of ReLUs) where it is written; the scheduler otherwise sinks them to just in front of their use.
        f32x4 h[10], h2[10], y[5];
        auto bias10 = [&](f32x4 (&acc)[10], const f32x4* bq) {
#pragma unroll
            for (int n = 0; n < 10; ++n) acc[n] = bq[n];
        };
        auto bias5 = [&](f32x4 (&acc)[5], const f32x4* bq) {
#pragma unroll
            for (int n = 0; n < 5; ++n) acc[n] = bq[n];
        };
        auto relu10 = [&](f32x4 (&acc)[10]) {   // the 40 ReLUs of a layer in one run (RBATCH): interleaved with the next layer's MFMAs,
            if constexpr (RBATCH) __builtin_amdgcn_sched_barrier(0);   // each would cost an s_nop for the VALU-write -> MFMA-read hazard
#if !(BNN_ABLATE & 1)
#pragma unroll
            for (int n = 0; n < 10; ++n) acc[n] = relu4(acc[n]);
#endif
            if constexpr (RBATCH) __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (PREF) bias10(h, bq1);
        // one tile of feature_nn: rows of tile `it` (in xv) -> latents y; fetches tile `it_next`'s rows into xv behind feature_nn.0
        auto tile = [&](const int it, const int it_next) {
            if constexpr (MEGNO) {   // summarize_megno (:480-484): Welford over the lane's timesteps, like the latents below
                const float rcm = p.rcp_tab[it];
                const float dl = xmeg - gmean;
                const float mn = fmaf(dl, rcm, gmean);
                gm2 = fmaf(dl, xmeg - mn, gm2);
                gmean = mn;
            }
            // feature_nn.0 + ReLU: MFMA m = k * 10 + n multiplies input column k into neuron group n (bias first, then the inputs in
            // ascending order: the oracle's natural order); its A operand is lanes 4(m&15).. of weight register m >> 4.
            {
                if constexpr (PREF) {
                    bias10(h2, bq2);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    bias10(h, bq1);
                }
                // masks, then add_input_noise (:486-504): masked columns become pure noise.  This lane's row is timestep
                // 4*it + ph0 of system sysc0; its 41 normals are the 7 Philox blocks t*7 + 0..6, six normals each (or the explicit
                // tensor's row).  A block is generated right in front of the six columns that consume it, so that its
                // registers are short-lived (the whole row's noise up front cost 13 spilled VGPRs).
                const float* er = nullptr;
                int tblk = 0;
                if constexpr (NOISY) {
                    const int t = 4 * it + ph0;
                    tblk = t * NIN_BLOCKS;
                    if constexpr (XNOISE) er = p.eps_in + (r * p.B + sysc0) * rowstride + (int64_t)t * F;
                }
                auto noise6 = [&](int blk) {
                    float n6[6];
                    if constexpr (XNOISE) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) n6[j] = (6 * blk + j < F) ? er[6 * blk + j] : 0.0f;
                    } else {
                        philox_in6(p.row_id0 + r, p.sys_id0 + sysc0, tblk + blk, p.seed, n6);
                    }
                    const f32x4* nb = reinterpret_cast<const f32x4*>(nsc + 96 + 8 * blk);
                    const f32x4 s0 = nb[0], s1 = nb[1], k0 = nb[14], k1 = nb[15];  // scales, keep-factors (56 floats further on)
                    // (indexed access only: __builtin_bit_cast of a swizzle member such as k0.y was miscompiled by hipcc 7.2 -- every
                    // column got k0.x's mask)
                    float scs[6], kpf[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) { scs[j] = j < 4 ? s0[j] : s1[j - 4]; kpf[j] = j < 4 ? k0[j] : k1[j - 4]; }
#pragma unroll
                    for (int j = 0; j < 6; j += 2) {   // column pairs: ONE v_pk_mul_f32 (randn * exp(logvar / 2)) and ONE v_pk_fma_f32
                        const int col = 6 * blk + j;   // x * keep + noise with keep = 1.0 | 0.0: x * 1.0 and x * 0.0 are exact, so the fma rounds once,
                        if (col + 1 < KIN) {           // exactly where the reference's x + noise does (:445) -- and the two v_and of a bit-mask form are gone
                            const f32x2 xp = {xv[col], xv[col + 1]};
                            const f32x2 kp = {kpf[j], kpf[j + 1]};
                            f32x2 nz = {n6[j], n6[j + 1]};
                            const f32x2 sc = {scs[j], scs[j + 1]};
                            nz = nz * sc;                                             // a multiply ...
                            const f32x2 xs = __builtin_elementwise_fma(xp, kp, nz);   // ... then the add
                            xv[col] = xs.x;
                            xv[col + 1] = xs.y;
                        } else if (col < KIN) {           // the last column stands alone (KIN is odd)
                            xv[col] = fmaf(xv[col], kpf[j], n6[j] * scs[j]);
                        }
                    }
                };
#if BNN_NIN16
                // measurement build (bnn_common.hip.h, BNN_NIN16): blocks of EIGHT normals, six per row, in-kernel Philox form only
                auto noise8 = [&](auto BLK) {
                    constexpr int blk = BLK;
                    constexpr int npairs = (F - 8 * blk + 1) / 2 < 4 ? (F - 8 * blk + 1) / 2 : 4;
                    float n8[8];
                    philox_in8<npairs>(p.row_id0 + r, p.sys_id0 + sysc0, (4 * it + ph0) * NIN16_BLOCKS + blk, p.seed, n8);
                    const f32x4* nb = reinterpret_cast<const f32x4*>(nsc + 96 + 8 * blk);
                    const f32x4 s0 = nb[0], s1 = nb[1], k0 = nb[14], k1 = nb[15];
                    float scs[8], kpf[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { scs[j] = j < 4 ? s0[j] : s1[j - 4]; kpf[j] = j < 4 ? k0[j] : k1[j - 4]; }
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const int col = 8 * blk + j;
                        if (col + 1 < KIN) {
                            const f32x2 xp = {xv[col], xv[col + 1]};
                            const f32x2 kp = {kpf[j], kpf[j + 1]};
                            f32x2 nz = {n8[j], n8[j + 1]};
                            const f32x2 sc = {scs[j], scs[j + 1]};
                            nz = nz * sc;
                            const f32x2 xs = __builtin_elementwise_fma(xp, kp, nz);
                            xv[col] = xs.x;
                            xv[col + 1] = xs.y;
                        } else if (col < KIN) {
                            xv[col] = fmaf(xv[col], kpf[j], n8[j] * scs[j]);
                        }
                    }
                };
#endif
                static_for<KIN * 10>([&](auto M) {
                    constexpr int m = M, k = m / 10, n = m % 10;
                    if constexpr (NOISY && BNN_NIN16 && !XNOISE) {
#if BNN_NIN16
                        if constexpr (n == 0 && k % 8 == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            noise8(std::integral_constant<int, k / 8>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
#endif
                    } else if constexpr (NOISY && n == 0 && k % 6 == 0) {  // a scheduling region of its own: the MFMAs around it do not move across
                        __builtin_amdgcn_sched_barrier(0);
                        noise6(k / 6);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    h[n] = mfma4b<(m & 15)>(wr[m >> 4], xv[k], h[n]);
                });
            }
            relu10(h);
#if !(BNN_ABLATE & 4)
            // x of this tile is dead: fetch the next tile's rows into the same registers, one tile of work to land.  All nine loads in
            // ONE burst: spread over the MFMAs of feature_nn.2 (one per 20 or 36) the kernel measured 14-20 % SLOWER
            // (profiles/r03_ab_variants.txt).
            load_row<KIN>(rowp + (int64_t)it_next * 4 * F, xv);
            if constexpr (MEGNO) xmeg = rowp[(int64_t)it_next * 4 * F + MEGNO_COL];
            asm volatile("" ::: "memory");
#endif
            // feature_nn.2 + ReLU: MFMA m = k * 10 + n
            if constexpr (!PREF) bias10(h2, bq2);
            static_for<H * 10>([&](auto M) {
                constexpr int m = M, k = m / 10, n = m % 10;
                h2[n] = mfma4b<(m & 15)>(wr[WRL::R1 + (m >> 4)], h[k >> 2][k & 3], h2[n]);
            });
            if constexpr (PREF) {
                __builtin_amdgcn_sched_barrier(0);
                bias5(y, bq3);   // lands while the ReLUs below issue
            }
            relu10(h2);
            // feature_nn.4: MFMA m = k * 5 + n
            if constexpr (PREF) {
                bias10(h, bq1);  // the NEXT tile's feature_nn.0 accumulators (h is dead from here on)
                __builtin_amdgcn_sched_barrier(0);
            } else {
                bias5(y, bq3);
            }
            static_for<H * 5>([&](auto M) {
                constexpr int m = M, k = m / 5, n = m % 5;
                y[n] = mfma4b<(m & 15)>(wr[WRL::R1 + WRL::R2 + (m >> 4)], h2[k >> 2][k & 3], y[n]);
            });
        };
        // torch.mean / torch.std over time (:418-419): Welford over this lane's timesteps; y = the latents of tile `it`, its (it + 1)-th.
        // The plain loop below carries the SAME statements inline: called through this lambda the SLP vectoriser splits the headline
        // loop's 20 packed subtractions into 20 scalar + 10 packed ones -- 10 more vector instructions per 910-MFMA tile, 2.4 % of the
        // kernel's cycles (round 6: seen in SQ_INSTS_VALU, 2.80e10 -> 2.96e10 per configs[2] launch, before it was seen in time).
        auto pool = [&](const int it) {
#if BNN_ABLATE & 2
#pragma unroll
            for (int n = 0; n < 5; ++n) asm volatile("" : : "v"(y[n]));   // the latents stay computed, nothing consumes them
#else
            const float rcn = p.rcp_tab[it];
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
        };
        if constexpr (!TSPLIT) {
            for (int it = 0; it < ntiles; ++it) {
                tile(it, (it + 1 < ntiles) ? it + 1 : it);
#if BNN_ABLATE & 2
                pool(it);
#else
                const float rcn = p.rcp_tab[it];
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
            }
        } else {
            for (int it0 = 0; it0 < ntiles; it0 += 4) {   // a round: tile it0 + w on wave w, then wave 0 pools the round's tiles in order
                const int it = it0 + wave;
                f32x4* yb = ybuf + ((it0 >> 2) & 1) * (4 * 5 * 64);
                if (it < ntiles) {
                    tile(it, (it + 4 < ntiles) ? it + 4 : it);
#pragma unroll
                    for (int n = 0; n < 5; ++n) yb[(wave * 5 + n) * 64 + lane] = y[n];
                }
                __syncthreads();   // (the next round writes the OTHER buffer; wave 0 is through with this one before it reaches the next barrier)
                if (wave == 0) {
                    for (int w = 0; w < 4 && it0 + w < ntiles; ++w) {
#pragma unroll
                        for (int n = 0; n < 5; ++n) y[n] = yb[(w * 5 + n) * 64 + lane];   // (its own tile's latents are in the buffer too: y is free)
                        pool(it0 + w);
                    }
                }
            }
            if (wave != 0) continue;   // the tail (merge, sampled moments, regress_nn, outputs) is wave 0's
        }
#if BNN_ABLATE & 8
        {
#pragma unroll
            for (int n = 0; n < 5; ++n) { asm volatile("" : : "v"(mean[n])); asm volatile("" : : "v"(m2[n])); }
            if (valid0 && ph0 == 0) p.out[(r * p.B + sys0) * 2] = mean[0][0];
            continue;
        }
#endif

        // Everything below the tile loop works from coordinates RE-derived here from a laundered copy of the lane id, so that
        // none of them (system ids, validity, pointers) is kept in a register -- or spilled -- across the loop.
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        const int g = lane_t >> 4, c = lane_t & 15;   // regress_nn (16x16x4) coordinates
        const int sl = lane_t >> 2, ph = lane_t & 3;  // feature_nn coordinates again
        const int64_t sys = wb0 + sl;
        const bool valid = sys < b1;
        const int64_t sysc = valid ? sys : b1 - 1;

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
        if constexpr (MEGNO) {   // the same two merge steps for the MEGNO column; then mean and torch.std (unbiased) -> LDS, summary
            float half_n = half_n0;
            {
                float om = quad_perm<0xB1>(gmean), o2 = quad_perm<0xB1>(gm2);
                float dl = om - gmean;
                float mm = (gmean + om) * 0.5f;
                gm2 = (gm2 + o2) + (dl * dl) * half_n;
                gmean = mm;
            }
            half_n = half_n * 2.0f;
            {
                float om = quad_perm<0x4E>(gmean), o2 = quad_perm<0x4E>(gm2);
                float dl = om - gmean;
                float mm = (gmean + om) * 0.5f;
                gm2 = (gm2 + o2) + (dl * dl) * half_n;
                gmean = mm;
            }
            const float gstd = sqrtf(gm2 / nm1);
            if (ph == 0) {
                float* mg = megscr + wave * MEGSCR + sl * 2;
                mg[0] = gmean;
                mg[1] = gstd;
                if (p.summary && valid) {
                    float* sp = p.summary + (r * p.B + sys) * Y::SM + S2;
                    sp[0] = gmean;
                    sp[1] = gstd;
                }
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
            float mu_s, sd_s;
            sampled_moments(mymean[j], mym2[j], e1[j], e2[j], nm1, nT, mu_s, sd_s);
            sumscr[sl * S2 + 5 * ph + j] = mu_s;
            sumscr[sl * S2 + L + 5 * ph + j] = sd_s;
            if (p.summary && valid) {
                float* sp = p.summary + (r * p.B + sys) * Y::SM + 5 * ph + j;
                sp[0] = mu_s;
                sp[L] = sd_s;
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- regress_nn on the 16 systems of this wave-batch (16x16x4 path): column c <-> system wb0 + c
        const int64_t sysb = wb0 + c;
        const bool validb = sysb < b1;
        float skeep[Y::NK4];
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) skeep[ks] = sumscr[c * S2 + kmap_summary(ks, g)];
        if constexpr (MEGNO) skeep[10] = g < 2 ? megscr[wave * MEGSCR + c * 2 + g] : 0.0f;   // torch.cat([summary_stats, megno_avg_std]) (:509-510)
        if constexpr (NOISY) {  // add_summary_noise (:448-450)
            const int64_t sc = validb ? sysb : b1 - 1;
            if constexpr (XNOISE) {
                const float* es = p.eps_sum + (r * p.B + sc) * Y::SM;
#pragma unroll
                for (int ks = 0; ks < 10; ++ks) skeep[ks] = skeep[ks] + es[kmap_summary(ks, g)] * nsc[F + kmap_summary(ks, g)];
                if constexpr (MEGNO)
                    if (g < 2) skeep[10] = skeep[10] + es[S2 + g] * nsc[F + S2 + g];
            } else {
                const int64_t grow = p.row_id0 + r, gsys = p.sys_id0 + sc;
#pragma unroll
                for (int kind = 0; kind < 2; ++kind) {
                    f32x4 a4n = philox_sys4(TAG_SUM, grow, gsys, kind * 5 + g, p.seed);
                    float bn = philox_sys4(TAG_SUM, grow, gsys, kind * 5 + 4, p.seed)[g];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        skeep[kind * 5 + rr] = skeep[kind * 5 + rr] + a4n[rr] * nsc[F + kmap_summary(kind * 5 + rr, g)];
                    skeep[kind * 5 + 4] = skeep[kind * 5 + 4] + bn * nsc[F + kmap_summary(kind * 5 + 4, g)];
                }
                if constexpr (MEGNO) {   // eps_sum[40 + g] = element g of quad 10
                    const f32x4 mn4 = philox_sys4(TAG_SUM, grow, gsys, 10, p.seed);
                    const float mnz = g == 0 ? mn4[0] : mn4[1];
                    if (g < 2) skeep[10] = skeep[10] + mnz * nsc[F + S2 + g];
                }
            }
        }
        const f32x4 a6 = regress16<MEGNO>(skeep, f2frag, lane);
        if (g == 0 && validb) {
            // predict_instability + soft_clamp (:295-296, :437-442)
            const float r0 = a6[0], r1 = a6[1];
            f32x2 ms = soft_clamp2(r0, r1, p.std_lo, p.std_span);
            if (bad_seed) ms.x = ms.y = __builtin_nanf("");
            if constexpr (STATS) {
                p.sink[r * p.B + sysb] = stats_draw(p.st, ms.x, ms.y, p.row_id0 + r, p.sys_id0 + sysb, p.seed);
            } else {
                const int64_t o = (r * p.B + sysb) * 2;
                *reinterpret_cast<f32x2*>(p.out + o) = ms;
                if (p.pre_clamp) *reinterpret_cast<f32x2*>(p.pre_clamp + o) = (f32x2){r0, r1};
            }
        }
        __builtin_amdgcn_wave_barrier();  // scratch is reused by the next wave-batch
    }
}

template <int KIN, bool FUSED, bool NOISY, bool STATS, bool MEGNO = false, bool XNOISE = false, bool TSPLIT = false>
inline hipError_t launch_forward_form(unsigned nblk, hipStream_t st, const FwdParams& p) {
    allow_big_lds<&bnn_forward_kernel<KIN, FUSED, NOISY, STATS, MEGNO, XNOISE, TSPLIT>>();   // once per (function, device), thread-safe
    constexpr size_t lds_bytes = fwd_lds_bytes<KIN, TSPLIT>();
    hipLaunchKernelGGL((bnn_forward_kernel<KIN, FUSED, NOISY, STATS, MEGNO, XNOISE, TSPLIT>), dim3(nblk), dim3(256), lds_bytes, st, p);
    return hipGetLastError();
}

}  // namespace bnn
