// bnn_generic.hip.h -- the GENERIC forward engine: VarModel.forward / forward_swag_fast for the network the
// reference builds from hparams (spock_reg_model.py:301-321, 346-362: any hidden / latent up to 128, depth `in` / `out`, 41 or 82
// features, fix_megno), any series length T >= 2 (:416-435), quiet or noisy (:444-450), optionally with the statistics tail.
// The pretrained ensemble's 41->40->40->20 / 40->40->40->2 network at T % 4 == 0 keeps its own kernel (bnn_forward.hip.h, weights
// register-resident); everything else the reference accepts runs here.
//
// Same decomposition as that kernel -- lane = row (lane l = system l >> 2 of the wave's 16, timestep 4 it + (l & 3)),
// v_mfma_f32_4x4x1_16b_f32 with the CBSZ/ABID broadcast, activations in registers from layer to layer, per-lane Welford over the
// lane's timesteps -- with the shapes taken from a descriptor (bnn_generic.h) instead of template constants:
//   * weight registers are streamed from an LDS image: one ds_read_b32 feeds 16 MFMAs (4 inputs x 4 neuron groups); a block's
//     registers are all requested up front, so the reads of a block are in flight together;
//   * a layer is a run-time loop over output blocks of 16 neurons; inside, the input quads are a compile-time loop with an early
//     exit (register arrays need compile-time indices), sized by the template bucket HQ = activation quads (48 / 64 / 96 / 128 wide);
//   * the pool state (mean, M2 per latent and lane) lives in LDS, so the latent width is a run-time number; lanes whose timestep
//     lies past T skip the update, and the four partitions (t & 3) are merged with their own counts (equal counts: the
//     symmetric form, i.e. the v50 kernel's bits; unequal: Chan's general form; constants from the host, as in the oracle);
//   * regress_nn runs through the same layer routine with lane = system (the four lanes of a quad repeat the work: 3 280 of
//     407 280 MACs for the v50 shapes), its weight registers from the LDS image when they fit, else gathered from the flat vector.
// Accumulation order per output: bias, then inputs ascending -- the oracle's natural order for every layer.
//
// ONE body, two ways to feed it its shapes: generic_body<FQ, HQ, W8, AS, NOISY> with AS = ArchRuntime is what the
// ahead-of-time buckets (bnn_fwd_generic*.hip) instantiate; a policy whose get() returns a constexpr GenArch -- written by
// bnn_spec_source for ONE network and compiled at run time (specialize.py), or generated ahead of time for the pretrained network
// (bnn_fwd_v50spec.hip) -- folds every count, and switches on (by its traits) the input-quad-major layer routine with one-step-ahead
// weight reads, the pool state in registers with a DPP merge, layer 0 over the unmasked columns only.  Same arithmetic, same order:
// bit-identical outputs.
#pragma once
#include "bnn_common.hip.h"
#include "bnn_generic.h"
#include "bnn_stats.hip.h"

namespace bnn {

// QMASK: bit q set = quad q of the row is read (specialised quiet forms leave out the quads whose columns are all masked: they are
// never multiplied -- layer 0 runs over the live columns only)
template <int FQ, uint32_t QMASK = 0xffffffffu>
DEVINL void gen_load_row(const float* __restrict__ rp, f32x4 (&xr)[FQ]) {
    static_assert(FQ == 11 || FQ == 21, "41 or 82 features");
    constexpr int NFULL = FQ - 1;
#pragma unroll
    for (int q = 0; q < NFULL; ++q) {
        if ((QMASK >> q) & 1u) xr[q] = *reinterpret_cast<const f32x4u*>(rp + 4 * q);
        else xr[q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    if constexpr (((QMASK >> NFULL) & 1u) == 0) {
        xr[NFULL] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    } else if constexpr (FQ == 11) {
        xr[10] = (f32x4){rp[40], 0.0f, 0.0f, 0.0f};
    } else {
        const f32x2 t = *reinterpret_cast<const f32x2u*>(rp + 80);
        xr[20] = (f32x4){t.x, t.y, 0.0f, 0.0f};
    }
}

// nn.ReLU (lim = 0) or identity (lim = INT_MIN) as one integer max on the bit pattern.  The element goes through a scalar
// parameter first: __builtin_bit_cast applied to a vector element directly reads element 0 (hipcc 7.2; bnn_forward.hip.h has the
// same note) -- the first version of this routine replicated neuron 4g of every group.
DEVINL float relu_lim1(float v, int lim) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_elementwise_max(b, lim));   // v_max_i32 (a compare + select otherwise: three instructions per element with the accumulator read)
}
DEVINL f32x4 relu_lim4(f32x4 v, int lim) {
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = relu_lim1(v[i], lim);
    return o;
}
// regress_nn (once per 16 systems): torch.relu's treatment of non-finite values (relu_ieee, bnn_common.hip.h) -- a pool that overflowed on
// finite inputs hands regress_nn inf / NaN, and the reference's answer is NaN.  IEEE = false: feature_nn's hot loop, one v_max_i32.
template <bool IEEE>
DEVINL f32x4 relu_sel4(f32x4 v, int lim) {
    if constexpr (IEEE) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = lim == 0 ? relu_ieee(v[i]) : v[i];
        return o;
    } else {
        return relu_lim4(v, lim);
    }
}

// Register layout of an activation vector (so that every array index is a compile-time constant AND no select is ever needed):
// a layer writes its FULL output blocks (16 neurons = 4 quads each) to quads 4 nb .. 4 nb + 3 and its LAST block -- whatever its index,
// which is only known at run time -- to the last four quads of the array.  The next layer therefore reads its first
// in_nfull = 4 (nblk_prev - 1) input quads from in[0 ..] and the remaining ntail = nkq - in_nfull (<= 4) from in[NQI - 4 ..].  x rows and
// summaries are laid out naturally (in_nfull = nkq, ntail = 0).  Weight register kq of a block is the LOGICAL input quad kq either way.
//
// One output block: 4 kk x NG MFMAs per input quad -- input 4 kq + kk into neuron group q (bias first, then inputs ascending) --
// accumulating straight into the block's output registers out[BASE .. BASE + 3].  The weight registers are requested a chunk of up to 16
// at a time (one ds_read_b32 each, all of a chunk in flight together).
template <int NG, int BASE, bool IEEE, int NQI, int NQO>
DEVINL void gen_block(const f32x4 (&in)[NQI], f32x4 (&out)[NQO], int in_nfull, int ntail, const float* wp, const f32x4* bq, int lim) {
#pragma unroll
    for (int q = 0; q < 4; ++q) out[BASE + q] = bq[q];
    float wt[4];
    static_while<4>([&](auto J) {   // the tail quads' registers: logical quads in_nfull + j
        constexpr int j = J;
        if (j >= ntail) return false;
        wt[j] = wp[(in_nfull + j) * 64];
        return true;
    });
    constexpr int CH = NQI < 16 ? NQI : 16;
    static_while<(NQI + CH - 1) / CH>([&](auto C) {
        constexpr int k0 = C * CH;
        if (k0 >= in_nfull) return false;
        float wv[CH];
        static_while<CH>([&](auto J) {
            constexpr int kq = k0 + J;
            if (kq >= NQI || kq >= in_nfull) return false;
            wv[J] = wp[kq * 64];
            return true;
        });
        static_while<CH>([&](auto J) {
            constexpr int kq = k0 + J;
            if (kq >= NQI || kq >= in_nfull) return false;
            static_for<16>([&](auto A) {
                constexpr int a = A, kk = a >> 2, q = a & 3;
                if constexpr (q < NG) out[BASE + q] = mfma4b<a>(wv[J], in[kq < NQI ? kq : 0][kk], out[BASE + q]);
            });
            return true;
        });
        return true;
    });
    static_while<4>([&](auto J) {
        constexpr int j = J;
        if (j >= ntail) return false;
        static_for<16>([&](auto A) {
            constexpr int a = A, kk = a >> 2, q = a & 3;
            if constexpr (q < NG) out[BASE + q] = mfma4b<a>(wt[j], in[NQI - 4 + j][kk], out[BASE + q]);
        });
        return true;
    });
#pragma unroll
    for (int q = 0; q < 4; ++q) out[BASE + q] = relu_sel4<IEEE>(out[BASE + q], lim);
}

// One Linear (+ ReLU) for the wave's 64 rows; weight register (nb, kq) = LDS image entry [(wreg0 + nb * nkq + kq) * 64 + lane].
// in_nfull: see the layout note above; the output's is 4 (nblk - 1).
// TRIM: the last block issues only its live neuron groups (feature_nn: the hot loop); without it the padded groups are multiplied
// by zero weights (regress_nn: once per 16 systems).
DEVINL void gen_stage_block(const GenLayer ly, int nb, const float* __restrict__ We, float* stage, int lane);
template <int NQI, int NQO, bool TRIM, bool STAGED = false>
DEVINL void gen_layer(const f32x4 (&in)[NQI], f32x4 (&out)[NQO], const GenLayer ly, int in_nfull, const float* wimg, const float* bimg,
                      int lane, const float* __restrict__ We = nullptr, float* stage = nullptr) {
    const int nkq = ly.nkq, nblk = ly.nblk, ntail = nkq - in_nfull;
    const int lim = ly.relu ? 0 : (int)0x80000000;
    const f32x4* bq = reinterpret_cast<const f32x4*>(bimg + ly.bias0);
    auto wsrc = [&](int nb) -> const float* {
        if constexpr (STAGED) {
            if (ly.wreg0 < 0) {   // not in the LDS image: this block's registers come through the wave's staging area
                __builtin_amdgcn_wave_barrier();
                gen_stage_block(ly, nb, We, stage, lane);
                __builtin_amdgcn_wave_barrier();
                return stage + lane;
            }
        }
        return wimg + (size_t)(ly.wreg0 + nb * nkq) * 64 + lane;
    };
    static_while<NQO / 4 - 1>([&](auto NB) {   // the full blocks, at their natural quads
        constexpr int nb = NB;
        if (nb >= nblk - 1) return false;
        gen_block<4, 4 * nb, !TRIM>(in, out, in_nfull, ntail, wsrc(nb), bq + 4 * nb, lim);
        return true;
    });
    const float* wl = wsrc(nblk - 1);          // the last block, at the array's last four quads
    const f32x4* bl = bq + 4 * (nblk - 1);
    const int ng = TRIM ? ly.ng_last : 4;
    if (ng == 4) gen_block<4, NQO - 4, !TRIM>(in, out, in_nfull, ntail, wl, bl, lim);
    else if (ng == 3) gen_block<3, NQO - 4, !TRIM>(in, out, in_nfull, ntail, wl, bl, lim);
    else if (ng == 2) gen_block<2, NQO - 4, !TRIM>(in, out, in_nfull, ntail, wl, bl, lim);
    else gen_block<1, NQO - 4, !TRIM>(in, out, in_nfull, ntail, wl, bl, lim);
}

// The same Linear, INPUT-QUAD-MAJOR (the specialised forms: every count below is a compile-time constant there, the early exits fold
// away and a layer is one straight-line run of MFMAs).  Step kq multiplies input quad kq into EVERY output block -- per output the
// order is still bias, then inputs ascending -- so that
//   * a step's 4 kk x (4 (nblk - 1) + ng_last) MFMAs interleave all of the layer's accumulator chains: consecutive MFMAs on one
//     accumulator are a whole row of neuron groups apart (the block-major order leaves the last block's 1-3 chains back to back,
//     an s_nop each), and
//   * the weight registers are consumed in image order (register (kq, nb) = entry kq * nblk + nb of the layer: AS::kq_major images)
//     one step AHEAD: step kq + 1's nblk registers are requested before step kq's MFMAs are issued, the first step's by the previous
//     layer (w0), and the tail steps' at the top of the layer -- no MFMA waits on an LDS read issued just above it.
// A scheduling barrier behind each request keeps the compiler from sinking the reads back down to their first use (it does,
// to shorten live ranges: the first specialised build waited lgkmcnt(0) in front of every 32 MFMAs).
#ifndef BNN_GEN_SCHED_MASK
#define BNN_GEN_SCHED_MASK 0
#endif
// BNN_GEN_ABLATE (measurement builds of the specialised forms only, scripts/spec_ablate_r04.sh; results are WRONG): 1 = accumulators start
// at zero instead of the bias image, 2 = no pool update, 4 = the first tile's x rows are reused, 8 = no ReLU, 16 = no weight reads past
// the first step of a layer
#ifndef BNN_GEN_ABLATE
#define BNN_GEN_ABLATE 0
#endif
template <int NQI, int NQO, bool TRIM, int NQN = NQO>
DEVINL void gen_layer_kq(const f32x4 (&in)[NQI], f32x4 (&out)[NQO], const GenLayer ly, int in_nfull, const float* wimg, const float* bimg, int lane,
                         float (&w0)[NQO / 4], bool have_w0, const GenLayer* next) {
    constexpr int NB = NQO / 4;
    const int nkq = ly.nkq, nblk = ly.nblk, ntail = nkq - in_nfull;
    const int lim = ly.relu ? 0 : (int)0x80000000;
    const int ng = TRIM ? ly.ng_last : 4;
    const f32x4* bq = reinterpret_cast<const f32x4*>(bimg + ly.bias0);
    const float* wp = wimg + (size_t)ly.wreg0 * 64 + lane;
    auto load = [&](float (&w)[NB], const float* s, int nb_) {   // one step's registers: blocks 0 .. nb_ - 2 at their slots, the last at slot NB - 1
        static_while<NB - 1>([&](auto NBI) {
            constexpr int nb = NBI;
            if (nb >= nb_ - 1) return false;
            w[nb] = s[nb * 64];
            return true;
        });
        w[NB - 1] = s[(nb_ - 1) * 64];
    };
    float wt[4][NB], w[2][NB];
    static_while<4>([&](auto J) {   // the tail steps (logical quads in_nfull + j): a whole layer to land
        constexpr int j = J;
        if (j >= ntail) return false;
        load(wt[j], wp + (size_t)(in_nfull + j) * nblk * 64, nblk);
        return true;
    });
    if (!have_w0) load(w0, wp, nblk);
    static_while<NB - 1>([&](auto NBI) {
        constexpr int nb = NBI;
        if (nb >= nblk - 1) return false;
#pragma unroll
        for (int q = 0; q < 4; ++q) out[4 * nb + q] = (BNN_GEN_ABLATE & 1) ? (f32x4){0, 0, 0, 0} : bq[4 * nb + q];
        return true;
    });
#pragma unroll
    for (int q = 0; q < 4; ++q) out[NQO - 4 + q] = (BNN_GEN_ABLATE & 1) ? (f32x4){0, 0, 0, 0} : bq[4 * (nblk - 1) + q];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) w[0][nb] = w0[nb];
    auto step = [&](const float (&wr)[NB], const f32x4 xin) {
        static_for<4>([&](auto KK) {
            constexpr int kk = KK;
            static_while<NB - 1>([&](auto NBI) {
                constexpr int nb = NBI;
                if (nb >= nblk - 1) return false;
                static_for<4>([&](auto Q) {
                    constexpr int q = Q;
                    out[4 * nb + q] = mfma4b<4 * kk + q>(wr[nb], xin[kk], out[4 * nb + q]);
                });
                return true;
            });
            static_for<4>([&](auto Q) {
                constexpr int q = Q;
                if (q < ng) out[NQO - 4 + q] = mfma4b<4 * kk + q>(wr[NB - 1], xin[kk], out[NQO - 4 + q]);
            });
        });
    };
    const bool pre_next = next != nullptr;
    static_while<NQI>([&](auto S) {
        constexpr int s_ = S;
        if (s_ >= in_nfull) return false;
        if ((BNN_GEN_ABLATE & 16) && s_ + 1 < in_nfull) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) w[(s_ + 1) & 1][nb] = w[s_ & 1][nb];
        } else if (s_ + 1 < in_nfull) load(w[(s_ + 1) & 1], wp + (size_t)(s_ + 1) * nblk * 64, nblk);
        else if (pre_next && ntail == 0) load(w0, wimg + (size_t)next->wreg0 * 64 + lane, next->nblk);
        __builtin_amdgcn_sched_barrier(BNN_GEN_SCHED_MASK);
        step(w[s_ & 1], in[s_]);
        return true;
    });
    static_while<4>([&](auto J) {
        constexpr int j = J;
        if (j >= ntail) return false;
        if (pre_next && j + 1 == ntail) {
            load(w0, wimg + (size_t)next->wreg0 * 64 + lane, next->nblk);
            __builtin_amdgcn_sched_barrier(BNN_GEN_SCHED_MASK);
        }
        step(wt[j], in[NQI - 4 + j]);
        return true;
    });
    if (BNN_GEN_ABLATE & 8) return;
    static_while<NB - 1>([&](auto NBI) {
        constexpr int nb = NBI;
        if (nb >= nblk - 1) return false;
#pragma unroll
        for (int q = 0; q < 4; ++q) out[4 * nb + q] = relu_sel4<!TRIM>(out[4 * nb + q], lim);
        return true;
    });
#pragma unroll
    for (int q = 0; q < 4; ++q) out[NQO - 4 + q] = relu_sel4<!TRIM>(out[NQO - 4 + q], lim);
}

// feature_nn layer LI of a specialised form whose weight registers STAY IN VGPRs across the tiles (policies with n_wres > 0: networks
// whose feature_nn is a few dozen registers, like the pretrained one): wres[] is loaded from the image once per workgroup, and the
// policy's constexpr tables (nkq(l), nblk(l), ng_last(l), wreg0(l), bias0(l), relu(l)) make every index a compile-time constant.  Same
// step order as gen_layer_kq -- bias, inputs ascending per output -- without a single LDS read in the layer but the biases.
template <int NQI, int NQO, class AS, int LI, int NW>
DEVINL void gen_layer_res(const f32x4 (&in)[NQI], f32x4 (&out)[NQO], const float* bimg, const float (&wres)[NW]) {
    constexpr int nkq = AS::nkq(LI), nblk = AS::nblk(LI), ng = AS::ng_last(LI), wr0 = AS::wreg0(LI);
    constexpr int in_nfull = LI == 0 ? nkq : 4 * (AS::nblk(LI > 0 ? LI - 1 : 0) - 1), ntail = nkq - in_nfull;
    constexpr int lim = AS::relu(LI) ? 0 : (int)0x80000000;
    static_assert(nkq <= NQI && 4 * nblk <= NQO && ntail >= 0 && ntail <= 4 && wr0 + nkq * nblk <= NW, "layer tables");
    const f32x4* bq = reinterpret_cast<const f32x4*>(bimg + AS::bias0(LI));
    static_for<nblk - 1>([&](auto NBI) {
        constexpr int nb = NBI;
#pragma unroll
        for (int q = 0; q < 4; ++q) out[4 * nb + q] = bq[4 * nb + q];
    });
#pragma unroll
    for (int q = 0; q < 4; ++q) out[NQO - 4 + q] = bq[4 * (nblk - 1) + q];
    static_for<nkq>([&](auto S) {
        constexpr int s_ = S, phys = s_ < in_nfull ? s_ : NQI - 4 + (s_ - in_nfull);
        static_for<4>([&](auto KK) {
            constexpr int kk = KK;
            static_for<nblk - 1>([&](auto NBI) {
                constexpr int nb = NBI;
                static_for<4>([&](auto Q) {
                    constexpr int q = Q;
                    out[4 * nb + q] = mfma4b<4 * kk + q>(wres[wr0 + s_ * nblk + nb], in[phys][kk], out[4 * nb + q]);
                });
            });
            static_for<ng>([&](auto Q) {
                constexpr int q = Q;
                out[NQO - 4 + q] = mfma4b<4 * kk + q>(wres[wr0 + s_ * nblk + nblk - 1], in[phys][kk], out[NQO - 4 + q]);
            });
        });
    });
    static_for<nblk - 1>([&](auto NBI) {
        constexpr int nb = NBI;
#pragma unroll
        for (int q = 0; q < 4; ++q) out[4 * nb + q] = relu_lim4(out[4 * nb + q], lim);
    });
#pragma unroll
    for (int q = 0; q < 4; ++q) out[NQO - 4 + q] = relu_lim4(out[NQO - 4 + q], lim);
}

// A regress_nn layer whose registers did not fit the LDS image: block nb's registers are gathered from the flat vector (L2) into the
// wave's staging area, eight loads in flight, and the layer routine above then runs one block at a time from there.
DEVINL void gen_stage_block(const GenLayer ly, int nb, const float* __restrict__ We, float* stage, int lane) {
    const int a_ = lane >> 2, kk = a_ >> 2, q = a_ & 3, i = lane & 3;
    const int neuron = 16 * nb + 4 * q + i;
    const bool nlive = neuron < ly.N;
    const float* wrow = We + ly.off_w + (int64_t)(nlive ? neuron : 0) * ly.K;
#pragma unroll 8
    for (int kq = 0; kq < ly.nkq; ++kq) {
        const int k = 4 * kq + kk;
        const bool live = nlive && k < ly.K;
        const float v = wrow[live ? k : 0];
        stage[kq * 64 + lane] = live ? v : 0.0f;
    }
}

DEVINL void gen_merge(const GenMerge mg, float& ma, float& qa, float mb, float qb) {
    if (mg.mode == 2) return;
    if (mg.mode == 3) { ma = mb; qa = qb; return; }
    const float dl = mb - ma;
    if (mg.mode == 0) {
        const float mm = (ma + mb) * 0.5f;
        qa = (qa + qb) + (dl * dl) * mg.w1;
        ma = mm;
    } else {
        const float mm = fmaf(dl, mg.w1, ma);
        qa = (qa + qb) + (dl * dl) * mg.w2;
        ma = mm;
    }
}

// The same merge between two lanes of a quad (state in registers, partner's through DPP): `self_is_b` orders the operands (a = the lower
// partition), and the pair (0,1) / (2,3) step takes its constants per lane -- both results are formed and one is selected.
DEVINL void gen_merge_lanes(const GenMerge lo, const GenMerge hi, bool sel_hi, bool self_is_b, float& m, float& q, float om, float oq) {
    const float ma = self_is_b ? om : m, qa = self_is_b ? oq : q, mb = self_is_b ? m : om, qb = self_is_b ? q : oq;
    float m1 = ma, q1 = qa, m2 = ma, q2 = qa;
    gen_merge(lo, m1, q1, mb, qb);
    gen_merge(hi, m2, q2, mb, qb);
    m = sel_hi ? m2 : m1;
    q = sel_hi ? q2 : q1;
}

// Where the kernel reads the network's shapes from.  ArchRuntime: the descriptor in global memory (scalar loads: the ahead-of-time
// buckets, any network).  A policy whose get() returns a constexpr GenArch by value (written by bnn_spec_source, compiled at run time for ONE
// network: specialize.py) turns every early exit, trip count and image offset below into a constant: the layer loops unroll into one
// straight-line tile body that the compiler schedules as a whole, like the pretrained network's own kernel.
// pool_regs_of<AS>::lq: latent groups whose Welford state the tile loop keeps in registers (policies that declare pool_lq; else 0 = LDS)
template <class AS, class = void>
struct pool_regs_of { static constexpr int lq = 0; };
template <class AS>
struct pool_regs_of<AS, std::void_t<decltype(AS::pool_lq)>> { static constexpr int lq = AS::pool_lq; };

// in_compact_of<AS>::q: input quads of layer 0 when the policy drops the masked columns (declares in_q and live(k)); else 0
template <class AS, class = void>
struct in_compact_of { static constexpr int q = 0; };
template <class AS>
struct in_compact_of<AS, std::void_t<decltype(AS::in_q)>> { static constexpr int q = AS::in_q; };

// resident_of<AS>::n: feature_nn weight registers a policy keeps in VGPRs across the tiles (declares n_wres and the layer tables; else 0)
template <class AS, class = void>
struct resident_of { static constexpr int n = 0; };
template <class AS>
struct resident_of<AS, std::void_t<decltype(AS::n_wres)>> { static constexpr int n = AS::n_wres; };

// x_late_of<AS>::value: the tile's rows are read at the top of the tile instead of a tile ahead (policies that declare x_late)
template <class AS, class = void>
struct x_late_of { static constexpr bool value = false; };
template <class AS>
struct x_late_of<AS, std::void_t<decltype(AS::x_late)>> { static constexpr bool value = AS::x_late; };

// xq_mask_of<AS>::value: quads of an input row the kernel reads (policies that declare x_quads; else all)
template <class AS, class = void>
struct xq_mask_of { static constexpr uint32_t value = 0xffffffffu; };
template <class AS>
struct xq_mask_of<AS, std::void_t<decltype(AS::x_quads)>> { static constexpr uint32_t value = AS::x_quads; };

struct ArchRuntime {
    static constexpr bool kq_major = false;   // weight image block-major (register (nb, kq) at nb * nkq + kq), gen_layer
    static DEVINL const GenArch& get(const GenParams& P) { return *P.g; }
};

// W8: the form compiled for 256 registers and launched with EIGHT waves per workgroup (two per SIMD sharing one weight image: the
// partner wave's MFMAs fill this wave's LDS / memory waits -- a single wave per SIMD spent a quarter of its cycles parked in
// s_waitcnt).  It exists for the 41-feature buckets of 12 and 16 quads and is chosen by the host when eight waves' LDS fits
// (same-box A/B on a hidden-64 / latent-16 network: +13 %; where only four waves fit, the 256-register form's spills cost 5 %, so
// those shapes run the 512-register form at one wave per SIMD, as the wider buckets always do).
// NOISY: -1 = read noisy (one kernel serves forward(noisy_val=False) and (True)); 0 / 1 = fixed at compile time (specialised forms).
template <int FQ, int HQ, bool W8, class AS, int NOISY>
DEVINL void generic_body(const GenParams& P, float* lds) {
    const FwdParams& p = P.f;
    const GenArch& G = AS::get(P);   // (a temporary of constants in the specialised forms: its lifetime is the reference's)
    const bool noisy = NOISY < 0 ? P.noisy != 0 : NOISY != 0;
    constexpr int NBLK_IN = FQ == 11 ? 7 : 14;   // Philox blocks of six normals per input row (41 / 82 columns)
    constexpr int FCOLS = FQ == 11 ? 41 : 82;
    float* wimg = lds;
    float* bimg = wimg + gen_wimg_floats(G);
    float* nsc = bimg + G.nbias;   // exp(input_noise_logvar / 2) [4 * fq] | exp(summary_noise_logvar / 2) [4 * smq] | per noise block [8] scales | [8] keep-factors (1.0 | 0.0)
    float* nsc_sum = nsc + 4 * FQ;
    float* nsc_blk = nsc_sum + 4 * G.smq;
    float* wave0 = nsc + gen_nsc_floats(G);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int sl0 = lane >> 2, ph0 = lane & 3;
    const int F = G.F, L = G.L, SM = G.SM, lq = G.lq, smq = G.smq;

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

    // ---- prologue: the draw's flat parameter vector (global) -> weight-register image, bias image, noise scales (LDS)
    const float* __restrict__ We = p.W + (int64_t)e * G.d;
    const int nl = G.n_feat + G.n_reg;
    {
        const int a_ = lane >> 2, kk = a_ >> 2, q = a_ & 3, i = lane & 3;
#pragma unroll 4
        for (int R = wave; R < G.nwreg; R += nwaves) {
            int li = 0;
            for (int l = 1; l < nl; ++l)
                if (G.layer[l].wreg0 >= 0 && R >= G.layer[l].wreg0) li = l;
            const GenLayer ly = G.layer[li];
            const int rr = R - ly.wreg0;
            const int nb = AS::kq_major ? rr % ly.nblk : rr / ly.nkq, kq = AS::kq_major ? rr / ly.nblk : rr - nb * ly.nkq;
            const int neuron = 16 * nb + 4 * q + i, k = 4 * kq + kk;
            // zero_megno / zero_mmr / zero_nan / zero_eplusminus (:452-500) as zero weights on the masked input columns; the noisy
            // forward keeps them (masked columns carry pure noise)
            int kc = k, stride = ly.K;   // column of the flat weight row behind logical input k
            if constexpr (in_compact_of<AS>::q > 0) {
                if (li == 0) { stride = G.F; kc = AS::live(k < ly.K ? k : 0); }
            }
            const bool masked = li == 0 && !noisy && kc < 64 && ((p.zero_mask >> kc) & 1ull);
            const bool live = neuron < ly.N && k < ly.K && !masked;
            const float v = We[live ? ly.off_w + neuron * stride + kc : 0];
            wimg[R * 64 + lane] = live ? v : 0.0f;
        }
        if (wave == 0) wimg[G.nwreg * 64 + lane] = 0.0f;
        for (int j = tid; j < G.nbias; j += blockDim.x) {
            int li = 0;
            for (int l = 1; l < nl; ++l)
                if (j >= G.layer[l].bias0) li = l;
            const int n = j - G.layer[li].bias0;
            bimg[j] = n < G.layer[li].N ? We[G.layer[li].off_b + n] : 0.0f;
        }
        if (noisy) {   // exp(input_noise_logvar/2) (:445), exp(summary_noise_logvar/2) (:449)
            for (int j = tid; j < 4 * FQ; j += blockDim.x) nsc[j] = j < F ? expf(We[G.off_inlv + j] / 2.0f) : 0.0f;
            for (int j = tid; j < 4 * smq; j += blockDim.x) nsc_sum[j] = j < SM ? expf(We[G.off_sumlv + j] / 2.0f) : 0.0f;
            for (int j = tid; j < 8 * NBLK_IN; j += blockDim.x) {   // the input scales per noise block, and keep-factors of the unmasked columns
                const int col = NIN_PER_BLOCK * (j >> 3) + (j & 7);
                const bool live = (j & 7) < NIN_PER_BLOCK && col < F;
                nsc_blk[j] = live ? expf(We[G.off_inlv + col] / 2.0f) : 0.0f;
                nsc_blk[8 * NBLK_IN + j] = (live && !(col < 64 && ((p.zero_mask >> col) & 1ull))) ? 1.0f : 0.0f;
            }
        }
    }
    __syncthreads();
    // resident forms: feature_nn's weight registers leave the image once, here
    constexpr int NWR = resident_of<AS>::n;
    float wres[NWR > 0 ? NWR : 1];
    if constexpr (NWR > 0) {
#pragma unroll
        for (int R = 0; R < NWR; ++R) wres[R] = wimg[R * 64 + lane];
    }

    const int T = p.T, ntiles = p.ntiles;
    const float nm1 = (float)(T - 1), nT = (float)T;
    const int64_t rowstride = (int64_t)T * F;
    const int SMS = gen_sum_stride(G);
    float* poolm = wave0 + (size_t)wave * gen_wave_floats(G);   // [lq][64 lanes][4] running means   (no such rows when G.pool_lds == 0:
    float* poolq = poolm + lq * 256;                            // [lq][64][4] running M2              state in registers, merged by DPP)
    float* stage = poolm;                                       // [hq][64] weight registers of one block (regress_nn layers outside the image):
                                                                // the pool rows again -- they are dead once the partitions are merged
    float* sumscr = poolm + gen_pool_stage_floats(G);           // [16 systems][SMS] the pool normals (entries n, L + n), overwritten in place by
                                                                // the summaries: the lane that consumes normal n of a system writes summary n
    float* epsscr = sumscr;
    float* megscr = sumscr + 16 * SMS;                          // [64 lanes][2] MEGNO partitions
    f32x4* poolm4 = reinterpret_cast<f32x4*>(poolm);
    f32x4* poolq4 = reinterpret_cast<f32x4*>(poolq);

    for (int64_t wb0 = b0 + (int64_t)wave * 16; wb0 < b1; wb0 += (int64_t)nwaves * 16) {
        const int64_t sys0 = wb0 + sl0;
        const bool valid0 = sys0 < b1;
        const int64_t sysc0 = valid0 ? sys0 : b1 - 1;
        const float* sysp = p.x + sysc0 * rowstride;
        if constexpr (pool_regs_of<AS>::lq == 0) {
            for (int g = 0; g < lq; ++g) {
                poolm4[g * 64 + lane] = (f32x4){0, 0, 0, 0};
                poolq4[g * 64 + lane] = (f32x4){0, 0, 0, 0};
            }
        }
        float gmean = 0.0f, gm2 = 0.0f;
        // XPREF: the next tile's rows are fetched right behind layer 1 of the current one (a tile of work to land).  The widest
        // bucket has no registers to hold them across the other layers (two 128-register activation arrays): it loads at the top of
        // the tile and waits (about 1 us of a tile of 20 us or more at those widths).
        constexpr bool XPREF = HQ < 32 && !(W8 && HQ > 12) && !x_late_of<AS>::value;   // (the eight-wave form of the 16-quad bucket has no registers for it either;
                                                                                       // sixteen-wave forms: three partner waves cover the load instead)
        f32x4 xr[FQ];
        if constexpr (XPREF) {
            gen_load_row<FQ, xq_mask_of<AS>::value>(sysp + (int64_t)(ph0 < T ? ph0 : T - 1) * F, xr);
            asm volatile("" ::: "memory");
        }
        f32x4 a[HQ], b[HQ];
        float w0[HQ / 4];   // kq-major forms: the first step's weight registers of the NEXT layer, requested by the layer before it
        if constexpr (AS::kq_major) {
            const float* s0 = wimg + (size_t)G.layer[0].wreg0 * 64 + lane;
            static_while<HQ / 4 - 1>([&](auto NBI) {
                constexpr int nb = NBI;
                if (nb >= G.layer[0].nblk - 1) return false;
                w0[nb] = s0[nb * 64];
                return true;
            });
            w0[HQ / 4 - 1] = s0[(G.layer[0].nblk - 1) * 64];
        }
        // torch.mean / torch.std over time (:418-419): Welford over this lane's timesteps, state in LDS
        const int lat_nfull = 4 * (G.layer[G.n_feat - 1].nblk - 1);   // latent groups at their natural quads; the rest at the array's tail
        // (specialised forms with AS::pool_regs: the state stays in registers across the tiles -- latent groups are compile-time there --
        // and is written to the same LDS rows once, in front of the tail)
        constexpr int PLQ = pool_regs_of<AS>::lq;
        f32x4 pm[PLQ > 0 ? PLQ : 1], pq[PLQ > 0 ? PLQ : 1];
        if constexpr (PLQ > 0) {
#pragma unroll
            for (int g = 0; g < PLQ; ++g) { pm[g] = (f32x4){0, 0, 0, 0}; pq[g] = (f32x4){0, 0, 0, 0}; }
        }
        auto welford_step = [](const f32x4 y, float rcn, f32x4& mean, f32x4& m2) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dl = y[i] - mean[i];
                const float mn = fmaf(dl, rcn, mean[i]);
                m2[i] = fmaf(dl, y[i] - mn, m2[i]);
                mean[i] = mn;
            }
        };
        float* latrow = nullptr;   // this lane's row of the latents output (debug / side-effect output, bnn_feature_nn_f32)
        auto latents_out = [&](const f32x4 y, int g) __attribute__((always_inline)) {
            if (latrow) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * g + i < L) latrow[4 * g + i] = y[i];
            }
        };
        auto welford_out = [&](const f32x4 y, int g, float rcn) __attribute__((always_inline)) {
            f32x4 mean = poolm4[g * 64 + lane], m2 = poolq4[g * 64 + lane];
            welford_step(y, rcn, mean, m2);
            poolm4[g * 64 + lane] = mean;
            poolq4[g * 64 + lane] = m2;
            latents_out(y, g);
        };
        auto pool = [&](const f32x4 (&y)[HQ], float rcn) __attribute__((always_inline)) {
            if constexpr (PLQ > 0) {
                static_for<PLQ>([&](auto GI) {
                    constexpr int g = GI, phys = g < AS::lat_nfull ? g : HQ - 4 + (g - AS::lat_nfull);
                    welford_step(y[phys], rcn, pm[g], pq[g]);
                    latents_out(y[phys], g);
                });
            } else {
                static_while<HQ - 4>([&](auto GI) {
                    constexpr int g = GI;
                    if (g >= lat_nfull) return false;
                    welford_out(y[g], g, rcn);
                    return true;
                });
                static_while<4>([&](auto JI) {
                    constexpr int j = JI;
                    if (lat_nfull + j >= lq) return false;
                    welford_out(y[HQ - 4 + j], lat_nfull + j, rcn);
                    return true;
                });
            }
        };

        for (int it = 0; it < ntiles; ++it) {
            const int t = 4 * it + ph0;
            const bool tv = t < T;
            const int tc = tv ? t : T - 1;
            const float rcn = p.rcp_tab[it];
            latrow = (p.latents && valid0 && tv) ? p.latents + (((r * p.B + sys0) * T) + t) * (int64_t)L : nullptr;
            if constexpr (!XPREF) gen_load_row<FQ, xq_mask_of<AS>::value>(sysp + (int64_t)tc * F, xr);
            if (G.megno && tv) {   // summarize_megno (:480-484): the RAW column, before the masks and before any noise
                const float xm = xr[MEGNO_COL >> 2][MEGNO_COL & 3];
                const float dl = xm - gmean;
                const float mn = fmaf(dl, rcn, gmean);
                gm2 = fmaf(dl, xm - mn, gm2);
                gmean = mn;
            }
            if (noisy) {   // masks, then add_input_noise (:486-506): masked columns become pure noise
                const float* er = p.eps_in ? p.eps_in + ((r * p.B + sysc0) * T + tc) * (int64_t)F : nullptr;
                static_for<NBLK_IN>([&](auto BLK) {
                    constexpr int blk = BLK;
                    float n6[6];
                    if (er) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) n6[j] = (6 * blk + j < F) ? er[6 * blk + j] : 0.0f;
                    } else {
                        philox_in6(p.row_id0 + r, p.sys_id0 + sysc0, tc * NBLK_IN + blk, p.seed, n6);
                    }
                    const f32x4* nb4 = reinterpret_cast<const f32x4*>(nsc_blk + 8 * blk);
                    const f32x4 s0 = nb4[0], s1 = nb4[1], k0 = nb4[2 * NBLK_IN], k1 = nb4[2 * NBLK_IN + 1];
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int col = 6 * blk + j;
                        if (col < FCOLS) {
                            const float sc = j < 4 ? s0[j] : s1[j - 4], kp = j < 4 ? k0[j] : k1[j - 4];
                            const float nz = n6[j] * sc;                                     // randn * exp(logvar / 2): a multiply ...
                            xr[col >> 2][col & 3] = fmaf(xr[col >> 2][col & 3], kp, nz);     // ... then the add (:445): x * keep (1.0 | 0.0, exact) + noise
                        }
                    }
                });
            }
            auto layer_first = [&](const f32x4 (&in_)[FQ], f32x4 (&out_)[HQ]) __attribute__((always_inline)) {
                if constexpr (in_compact_of<AS>::q > 0) {   // the unmasked columns only: a renaming of the row's registers
                    constexpr int CQ = in_compact_of<AS>::q;
                    f32x4 xc[CQ];
                    static_for<CQ>([&](auto Q_) {
                        static_for<4>([&](auto KK) {
                            constexpr int q_ = Q_, kk = KK, c = AS::live(4 * q_ + kk);
                            xc[q_][kk] = in_[c >> 2][c & 3];
                        });
                    });
                    gen_layer_kq<CQ, HQ, true>(xc, out_, G.layer[0], G.layer[0].nkq, wimg, bimg, lane, w0, true, &G.layer[G.n_feat > 1 ? 1 : 0]);
                } else if constexpr (AS::kq_major) gen_layer_kq<FQ, HQ, true>(in_, out_, G.layer[0], G.layer[0].nkq, wimg, bimg, lane, w0, true, &G.layer[G.n_feat > 1 ? 1 : 0]);
                else gen_layer<FQ, HQ, true>(in_, out_, G.layer[0], G.layer[0].nkq, wimg, bimg, lane);
            };
            auto layer_next = [&](const f32x4 (&in_)[HQ], f32x4 (&out_)[HQ], int l_) __attribute__((always_inline)) {
                if constexpr (AS::kq_major)
                    gen_layer_kq<HQ, HQ, true>(in_, out_, G.layer[l_], 4 * (G.layer[l_ - 1].nblk - 1), wimg, bimg, lane, w0, true, &G.layer[l_ + 1 < G.n_feat ? l_ + 1 : 0]);
                else gen_layer<HQ, HQ, true>(in_, out_, G.layer[l_], 4 * (G.layer[l_ - 1].nblk - 1), wimg, bimg, lane);
            };
            if constexpr (NWR > 0) {   // weights in registers: every layer index is a compile-time constant (gen_layer_res)
                if constexpr (in_compact_of<AS>::q > 0) {
                    constexpr int CQ = in_compact_of<AS>::q;
                    f32x4 xc[CQ];
                    static_for<CQ>([&](auto Q_) {
                        static_for<4>([&](auto KK) {
                            constexpr int q_ = Q_, kk = KK, c = AS::live(4 * q_ + kk);
                            xc[q_][kk] = xr[c >> 2][c & 3];
                        });
                    });
                    gen_layer_res<CQ, HQ, AS, 0>(xc, a, bimg, wres);
                } else {
                    gen_layer_res<FQ, HQ, AS, 0>(xr, a, bimg, wres);
                }
            } else {
                layer_first(xr, a);
            }
            // x of this tile is dead: fetch the next tile's rows into the same registers
            if constexpr (XPREF && !(BNN_GEN_ABLATE & 4)) {
                const int tn = 4 * (it + 1) + ph0;
                gen_load_row<FQ, xq_mask_of<AS>::value>(sysp + (int64_t)(tn < T ? tn : T - 1) * F, xr);
                asm volatile("" ::: "memory");
            }
            if constexpr (NWR > 0) {
                static_for<AS::n_feat - 1>([&](auto LI_) {   // ping-pong between the two register arrays; layer l reads a for odd l
                    constexpr int l_ = LI_ + 1;
                    if constexpr (l_ % 2 == 1) gen_layer_res<HQ, HQ, AS, l_>(a, b, bimg, wres);
                    else gen_layer_res<HQ, HQ, AS, l_>(b, a, bimg, wres);
                });
                if (tv && !(BNN_GEN_ABLATE & 2)) {
                    if constexpr (AS::n_feat % 2 == 0) pool(b, rcn);
                    else pool(a, rcn);
                }
            } else {
            // the remaining Linear modules of feature_nn, ping-pong between the two register arrays
            int l = 1;
            for (; l + 1 < G.n_feat; l += 2) {
                layer_next(a, b, l);
                layer_next(b, a, l + 1);
            }
            if (l < G.n_feat) {
                layer_next(a, b, l);
                if (tv && !(BNN_GEN_ABLATE & 2)) pool(b, rcn);   // lanes past T sit the tile out
                if (BNN_GEN_ABLATE & 2) asm volatile("" :: "v"(b[0]), "v"(b[HQ - 4]));
            } else if (tv) {
                pool(a, rcn);
            }
            }
        }

        // ---- tail: merge the four partitions of every latent, sampled moments (compute_summary_stats :420-431)
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        const int sl = lane_t >> 2, ph = lane_t & 3;
        const int64_t sys = wb0 + sl;
        const bool valid = sys < b1;
        const int64_t sysc = valid ? sys : b1 - 1;
        const int64_t grow = p.row_id0 + r, gsys = p.sys_id0 + sysc;
        if (G.megno) { megscr[lane_t * 2] = gmean; megscr[lane_t * 2 + 1] = gm2; }
        if (!p.eps) {   // the system's 2 L pool normals: Philox blocks ph, ph + 4, ... of the quad's four lanes
            for (int qd = ph; 2 * qd < L; qd += 4) *reinterpret_cast<f32x4*>(epsscr + sl * SMS + 4 * qd) = philox_eps4(grow, gsys, qd, p.seed);
        }
        __builtin_amdgcn_wave_barrier();   // the pool state and the normals written above are read across lanes below (one wave's LDS
                                           // operations complete in order: no wait is needed, only the compiler must not reorder)
        auto finish = [&](int n, float mean_, float m2_) __attribute__((always_inline)) {   // latent n of this lane's system: sampled moments -> summary
            float e1, e2;
            if (p.eps) {
                const float* ep = p.eps + (r * p.B + sysc) * 2 * L;
                e1 = ep[n];
                e2 = ep[L + n];
            } else {
                e1 = epsscr[sl * SMS + n];
                e2 = epsscr[sl * SMS + L + n];
            }
            float mu_s, sd_s;
            sampled_moments(mean_, m2_, e1, e2, nm1, nT, mu_s, sd_s);
            sumscr[sl * SMS + n] = mu_s;
            sumscr[sl * SMS + L + n] = sd_s;
            if (p.summary && valid) {
                float* sp = p.summary + (r * p.B + sys) * SM;
                sp[n] = mu_s;
                sp[L + n] = sd_s;
            }
        };
        if constexpr (PLQ > 0) {
            // state in registers: the quad's four lanes hold the four partitions of the same latents -- (0,1) and (2,3) merge through
            // quad_perm [1,0,3,2], the halves through [2,3,0,1], every lane ends with the system's merged state; lane ph finishes latent 4 g + ph
            const bool odd = (ph & 1) != 0, upper = (ph & 2) != 0;
            static_for<PLQ>([&](auto GI) {
                constexpr int g = GI;
                f32x4 m = pm[g], q2 = pq[g];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float mi = m[i], qi = q2[i];
                    gen_merge_lanes(P.m01, P.m23, upper, odd, mi, qi, quad_perm<0xB1>(mi), quad_perm<0xB1>(qi));
                    gen_merge_lanes(P.m0123, P.m0123, false, upper, mi, qi, quad_perm<0x4E>(mi), quad_perm<0x4E>(qi));
                    m[i] = mi; q2[i] = qi;
                }
                const float mm = ph == 0 ? m[0] : ph == 1 ? m[1] : ph == 2 ? m[2] : m[3];
                const float qq = ph == 0 ? q2[0] : ph == 1 ? q2[1] : ph == 2 ? q2[2] : q2[3];
                if (4 * g + ph < L) finish(4 * g + ph, mm, qq);
            });
        } else {
            for (int n = ph; n < L; n += 4) {
                const int g = n >> 2, c = n & 3;
                float m[4], q2[4];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int idx = (g * 64 + 4 * sl + pp) * 4 + c;
                    m[pp] = poolm[idx];
                    q2[pp] = poolq[idx];
                }
                gen_merge(P.m01, m[0], q2[0], m[1], q2[1]);
                gen_merge(P.m23, m[2], q2[2], m[3], q2[3]);
                gen_merge(P.m0123, m[0], q2[0], m[2], q2[2]);
                finish(n, m[0], q2[0]);
            }
        }
        if (ph == 0) {
            if (G.megno) {   // torch.cat([summary_stats, megno_avg_std]) (:509-510): mean and unbiased std of the raw column
                float m[4], q2[4];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) { m[pp] = megscr[(4 * sl + pp) * 2]; q2[pp] = megscr[(4 * sl + pp) * 2 + 1]; }
                gen_merge(P.m01, m[0], q2[0], m[1], q2[1]);
                gen_merge(P.m23, m[2], q2[2], m[3], q2[3]);
                gen_merge(P.m0123, m[0], q2[0], m[2], q2[2]);
                const float gstd = sqrtf(q2[0] / nm1);
                sumscr[sl * SMS + 2 * L] = m[0];
                sumscr[sl * SMS + 2 * L + 1] = gstd;
                if (p.summary && valid) {
                    float* sp = p.summary + (r * p.B + sys) * SM + 2 * L;
                    sp[0] = m[0];
                    sp[1] = gstd;
                }
            }
            for (int n = SM; n < SMS; ++n) sumscr[sl * SMS + n] = 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
        if (noisy) {   // add_summary_noise (:448-450), on the LDS copy: lane ph takes summary quads ph, ph + 4, ...
            const float* es = p.eps_sum ? p.eps_sum + (r * p.B + sysc) * SM : nullptr;
            for (int kq = ph; kq < smq; kq += 4) {
                f32x4 nz;
                if (es) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) nz[j] = (4 * kq + j < SM) ? es[4 * kq + j] : 0.0f;
                } else {
                    nz = philox_sys4(TAG_SUM, grow, gsys, kq, p.seed);
                }
                const f32x4 sc = *reinterpret_cast<const f32x4*>(nsc_sum + 4 * kq);
                f32x4 sv = *reinterpret_cast<const f32x4*>(sumscr + sl * SMS + 4 * kq);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t2 = nz[j] * sc[j];
                    sv[j] = sv[j] + t2;
                }
                *reinterpret_cast<f32x4*>(sumscr + sl * SMS + 4 * kq) = sv;
            }
            __builtin_amdgcn_wave_barrier();
        }

        // ---- regress_nn + soft_clamp (predict_instability :437-442): lane = system (the quad's four lanes repeat the work)
        static_while<HQ>([&](auto KQ) {
            constexpr int kq = KQ;
            if (kq >= smq) return false;
            a[kq] = *reinterpret_cast<const f32x4*>(sumscr + sl * SMS + 4 * kq);
            return true;
        });
        float r0, r1;   // the two outputs are neurons 0, 1 of the last Linear's only (= last) block
        {
            int nfull = smq;   // the summary sits at its natural quads
            if constexpr (AS::kq_major) {   // (compile-time layer count: unrolled, every layer's shapes fold)
                static_for<AS::n_reg>([&](auto LI) {
                    constexpr int l = AS::n_feat + LI;
                    if (G.layer[l].wreg0 >= 0) gen_layer_kq<HQ, HQ, false>(a, b, G.layer[l], nfull, wimg, bimg, lane, w0, false, nullptr);
                    else gen_layer<HQ, HQ, false, true>(a, b, G.layer[l], nfull, wimg, bimg, lane, We, stage);
                    nfull = 4 * (G.layer[l].nblk - 1);
#pragma unroll
                    for (int i = 0; i < HQ; ++i) a[i] = b[i];
                });
            } else {
                for (int l = G.n_feat; l < nl; ++l) {
                    gen_layer<HQ, HQ, false, true>(a, b, G.layer[l], nfull, wimg, bimg, lane, We, stage);
                    nfull = 4 * (G.layer[l].nblk - 1);
#pragma unroll
                    for (int i = 0; i < HQ; ++i) a[i] = b[i];
                }
            }
            r0 = a[HQ - 4][0]; r1 = a[HQ - 4][1];
        }
        if (ph == 0 && valid) {
            const f32x2 ms = soft_clamp2(r0, r1, p.std_lo, p.std_span);
            if (p.sink) {
                p.sink[r * p.B + sys] = stats_draw(p.st, ms.x, ms.y, grow, p.sys_id0 + sys, p.seed);
            } else if (p.out) {   // (null for a latents-only call, bnn_feature_nn_f32)
                const int64_t o = (r * p.B + sys) * 2;
                *reinterpret_cast<f32x2*>(p.out + o) = ms;
                if (p.pre_clamp) *reinterpret_cast<f32x2*>(p.pre_clamp + o) = (f32x2){r0, r1};
            }
        }
        __builtin_amdgcn_wave_barrier();  // scratch is reused by the next wave-batch
    }
}

template <int FQ, int HQ, bool W8>
__global__ __launch_bounds__(W8 ? 512 : 256, 1) void bnn_forward_generic_kernel(const GenParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    generic_body<FQ, HQ, W8, ArchRuntime, -1>(P, lds);
}

template <int FQ, int HQ, bool W8>
inline hipError_t launch_generic_form(unsigned nblk, hipStream_t st, const GenParams& P, int nwaves, size_t lds_bytes) {
    allow_big_lds<&bnn_forward_generic_kernel<FQ, HQ, W8>>();   // once per (function, device), thread-safe
    hipLaunchKernelGGL((bnn_forward_generic_kernel<FQ, HQ, W8>), dim3(nblk), dim3(64 * nwaves), lds_bytes, st, P);
    return hipGetLastError();
}

}  // namespace bnn
