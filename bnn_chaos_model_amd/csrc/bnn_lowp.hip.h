// bnn_lowp.hip.h -- OPT-IN reduced-precision forward kernels for BASELINE.json configs[4] ("bf16 vs fp32 tolerance sweep"):
// feature_nn on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, fp32 accumulate), everything after the time pool (sampled
// moments, regress_nn on the exact-fp32 16x16x4 path, soft_clamp) exactly as in the fp32 kernel.  Never the default, never the
// headline: outputs differ from the reference by far more than the 1e-5 parity bar (the sweep in DESIGN.md says by how much).
//
//   NS = 1  "bf16":    x, weights, biases and activations rounded to bf16 (RNE), one product per layer
//   NS = 2  "bf16x3":  every operand split into hi + lo bf16 parts (16 significant bits), 3 products  hi*hi + hi*lo + lo*hi
//   NS = 3  "bf16x6":  three parts (24 significant bits = all of fp32), the 6 products of order <= 2: fp32-level error
//
// Data flow (D = W * X^T, so that a layer's accumulators ARE the next layer's B operand, no lane movement, no LDS):
//   C/D   lane (g = lane>>4, c = lane&15), register r of m-tile mt  = out[neuron 16 mt + 4 g + r][data row c]
//   B     lane (g, c), element j of k-step s                        = act[k(s, g, j)][data row c]
//   A     lane (g, m = lane&15), element j of k-step s              = W[neuron 16 mt + m][k(s, g, j)]
// The k order is free, so k(s, g, j) is chosen to be what the lane already holds:
//   layer 1 (31 live columns + bias = 32 slots, ONE k-step): group g reads 8 consecutive floats of its row at column 8 + 8g;
//            group 3 = columns 32..37, then column 0, then the constant 1.0 whose weight is the bias;
//   layers 2, 3 (40 inputs = two k-steps, the second half empty): s = 0: j < 4 -> neuron 4g + j (m-tile 0), j >= 4 -> neuron
//            16 + 4g + j - 4 (m-tile 1); s = 1: j < 4 -> neuron 32 + 4g + j (m-tile 2, neurons >= 40 are zero padding),
//            j = 4 -> the constant 1.0 (bias), j > 4 -> 0.
// A data row c = (system c>>2, timestep phase c&3): a tile is 4 systems x 4 timesteps, 25 tiles cover T = 100; the time pool is a
// per-lane Welford over the lane's 25 timesteps + a DPP merge over the 4 phases, as in the fp32 kernel.
// Weights live in registers for the whole workgroup (13 fragments x NS parts x 4 VGPRs).  13 (x1, x3, x6) MFMAs of 16 cycles per
// 16 rows against 1 820 cycles of fp32 MFMA: the bf16 forms are bound by their vector work (ReLU, splits, pool), not the pipe.
#pragma once
#include "bnn_common.hip.h"

namespace bnn {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// Element type of the matrix operands: H = false bfloat16 (8 significant bits, fp32's exponent range), H = true IEEE half (11
// significant bits, |v| < 65 504: values beyond that become inf).  Packing: {elem(b), elem(a)}, a in the low half, round to nearest
// even.  A vector conversion, NOT inline asm: hipcc emits one v_cvt_pk_* for it and -- unlike for an asm statement -- inserts the
// wait states a following MFMA needs before it reads the result (with the asm form the MFMAs read stale operands: NaNs).
template <bool H>
DEVINL uint32_t cvt_pk(float a, float b) {
    const f32x2 v = {a, b};
    if constexpr (H) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <bool H>
DEVINL float lo_as_f32(uint32_t pk) {
    if constexpr (H) return (float)__builtin_bit_cast(f16x2, pk).x;
    else return __builtin_bit_cast(float, pk << 16);
}
template <bool H>
DEVINL float hi_as_f32(uint32_t pk) {
    if constexpr (H) return (float)__builtin_bit_cast(f16x2, pk).y;
    else return __builtin_bit_cast(float, pk & 0xffff0000u);
}
template <bool H>
constexpr uint32_t ONE_LO = H ? 0x00003C00u : 0x00003F80u;  // {0, 1.0}: the constant 1.0 of a bias slot in the low half

// (a, b) -> NS packed parts; part p holds bf16 / half of what is left after parts < p (the subtractions are exact in fp32).
// This translation unit is compiled with -fno-slp-vectorize (build.py), so packed fp32 instructions come from explicit two-element
// vector arithmetic only -- and, for the half forms, the residual of a pair becomes TWO instructions: v_fma_mixlo_f16 /
// v_fma_mixhi_f16 compute RN_half(fma(float(hi), -1, a)) straight from the packed hi part (the fma's result a - hi is exact, so
// this is the same number as converting, subtracting and converting back: 3 instructions per pair instead of 5).  The compiler
// only selects the mix forms for an fma whose multiplier it cannot see, hence the laundered -1.0 (`m1`).
template <int NS, bool H>
DEVINL void split_pair(float a, float b, float m1, uint32_t (&out)[NS]) {
    if constexpr (H && NS == 2) {
        out[0] = cvt_pk<H>(a, b);
        const f16x2 hi = __builtin_bit_cast(f16x2, out[0]);
        const f16x2 lo = {(_Float16)__builtin_fmaf((float)hi.x, m1, a), (_Float16)__builtin_fmaf((float)hi.y, m1, b)};
        out[1] = __builtin_bit_cast(uint32_t, lo);
    } else {
        f32x2 v = {a, b};
#pragma unroll
        for (int p = 0; p < NS; ++p) {
            out[p] = cvt_pk<H>(v.x, v.y);
            if (p + 1 < NS) v = v - (f32x2){lo_as_f32<H>(out[p]), hi_as_f32<H>(out[p])};   // one v_pk_add_f32
        }
    }
}
DEVINL float laundered_minus_one() {
    float m1 = -1.0f;
    asm volatile("" : "+s"(m1));
    return m1;
}

template <int NS>
struct Frag {  // one k-step of one operand: NS parts of 8 bf16
    u32x4 part[NS];
};

template <bool H>
DEVINL f32x4 mfma16(const u32x4& a, const u32x4& b, f32x4 c) {
    if constexpr (H) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// sum over the products of order <= NS - 1, smallest terms first
template <int NS, bool H>
DEVINL f32x4 mfma_split(const Frag<NS>& a, const Frag<NS>& b, f32x4 c) {
#pragma unroll
    for (int ord = NS - 1; ord >= 0; --ord)
#pragma unroll
        for (int i = 0; i <= ord; ++i) c = mfma16<H>(a.part[i], b.part[ord - i], c);
    return c;
}

// index into the flat parameter vector (or ZERO_IDX) of the weight that A-fragment element j of lane (g, m) holds
DEVINL int lowp_widx(int layer, int mt, int s, int g, int m, int j) {
    const int n = 16 * mt + m;
    if (layer == 0) {
        if (n >= H) return ZERO_IDX;
        if (g < 3) return OFF_W1 + n * F + 8 + 8 * g + j;
        if (j < 6) return OFF_W1 + n * F + 32 + j;
        return j == 6 ? OFF_W1 + n * F : OFF_B1 + n;
    }
    const int n_out = layer == 1 ? H : L;
    if (n >= n_out) return ZERO_IDX;
    const int off_w = layer == 1 ? OFF_W2 : OFF_W3, off_b = layer == 1 ? OFF_B2 : OFF_B3;
    int k;
    if (s == 0) k = j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);
    else {
        if (j == 4) return g == 0 ? off_b + n : ZERO_IDX;
        if (j > 4) return ZERO_IDX;
        k = 32 + 4 * g + j;
        if (k >= H) return ZERO_IDX;
    }
    return off_w + n * H + k;
}

template <int NS, bool H>
DEVINL Frag<NS> lowp_wfrag(const float* flat, int layer, int mt, int s, int g, int m, float m1) {
    Frag<NS> f;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        uint32_t pk[NS];
        split_pair<NS, H>(flat[lowp_widx(layer, mt, s, g, m, 2 * jp)], flat[lowp_widx(layer, mt, s, g, m, 2 * jp + 1)], m1, pk);
#pragma unroll
        for (int p = 0; p < NS; ++p) f.part[p][jp] = pk[p];
    }
    return f;
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
DEVINL uint32_t relu_pk16(uint32_t pk) {  // ReLU on two packed bf16 / half: one v_pk_max_i16 (negative floats are negative int16s)
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), z));
}

// ReLU + split of a layer's three accumulator tiles into the next layer's two B k-steps
template <int NS, bool H>
DEVINL void lowp_next_operand(const f32x4 (&acc)[3], float m1, Frag<NS>& b0, Frag<NS>& b1) {
    if constexpr (NS == 1) {  // round first, then ReLU on the packed pairs: bf16(relu(v)) == relu(bf16(v)), half the instructions
        b0.part[0][0] = relu_pk16(cvt_pk<H>(acc[0][0], acc[0][1]));
        b0.part[0][1] = relu_pk16(cvt_pk<H>(acc[0][2], acc[0][3]));
        b0.part[0][2] = relu_pk16(cvt_pk<H>(acc[1][0], acc[1][1]));
        b0.part[0][3] = relu_pk16(cvt_pk<H>(acc[1][2], acc[1][3]));
        b1.part[0][0] = relu_pk16(cvt_pk<H>(acc[2][0], acc[2][1]));
        b1.part[0][1] = relu_pk16(cvt_pk<H>(acc[2][2], acc[2][3]));
        return;  // elements 4..7 of b1 (the constant 1.0 of the bias slot, zeros) are set once by the caller
    }
    f32x4 t0 = relu4(acc[0]), t1 = relu4(acc[1]), t2 = relu4(acc[2]);
    uint32_t pk[NS];
    split_pair<NS, H>(t0[0], t0[1], m1, pk);
#pragma unroll
    for (int p = 0; p < NS; ++p) b0.part[p][0] = pk[p];
    split_pair<NS, H>(t0[2], t0[3], m1, pk);
#pragma unroll
    for (int p = 0; p < NS; ++p) b0.part[p][1] = pk[p];
    split_pair<NS, H>(t1[0], t1[1], m1, pk);
#pragma unroll
    for (int p = 0; p < NS; ++p) b0.part[p][2] = pk[p];
    split_pair<NS, H>(t1[2], t1[3], m1, pk);
#pragma unroll
    for (int p = 0; p < NS; ++p) b0.part[p][3] = pk[p];
    split_pair<NS, H>(t2[0], t2[1], m1, pk);
#pragma unroll
    for (int p = 0; p < NS; ++p) b1.part[p][0] = pk[p];
    split_pair<NS, H>(t2[2], t2[3], m1, pk);
#pragma unroll
    for (int p = 0; p < NS; ++p) b1.part[p][1] = pk[p];
}

// per wave: [XST] x-tile staging buffer (8 chunks x 16 rows x 16 B = 512 floats; doubles as the Philox scratch of the finish, which
// needs 16 * S2 = 640) + [16 * S2] summaries of the wave-batch
constexpr int XST = 16 * S2;                       // 640 floats
constexpr int SCRL = XST + 16 * S2;
// staged tile, chunk-major with an XOR swizzle: 16-byte slot of (chunk k, row r) = 16 k + (r ^ (k & 3))
DEVINL int xslot(int k, int r) { return 16 * k + (r ^ (k & 3)); }
constexpr size_t lowp_lds_bytes() { return sizeof(float) * (FLAT_LDS + 4 * SCRL); }

template <int NS, bool H>
__global__ __launch_bounds__(256, 2) void bnn_forward_lowp_kernel(const FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if constexpr (H) {
        // MODE.FP16_OVFL = 1: conversions to half saturate at +-65 504 instead of producing inf (hwreg MODE = id 1, bit 23, 1 bit).
        // Values out of half's range are then WRONG but finite -- e.g. the scripts' constant-4 fill of unstable systems
        // (figures/multiswag_5_planet.py:215) standardises the three mass columns to 1.9e5; their outputs are discarded there.
        __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);
    }
    float* flat = lds;            // [FLAT_LDS] flat parameter vector + zero slot, later ...
    float* f2frag = lds;          // ... [NF2][64] regress_nn operands in fragment order
    float* scr = lds + FLAT_LDS;  // [4][SCRL] per-wave scratch: x-tile staging / Philox normals, summaries, the constant 1.0

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int sq = c >> 2, tph = c & 3;  // system within the tile, timestep phase

    const WorkItem wi = work_item(p);
    const int e = wi.e;
    const int64_t sub = wi.sub;
    const int ch = e % p.nch;
    const int64_t r = e / p.nch;
    // the draw's chunk of systems (torch.chunk over the cB systems of the WHOLE batch, of which this call holds rows [coff, coff + B))
    const int64_t g0 = (int64_t)ch * p.csz - p.coff, gend = p.cB - p.coff;
    const int64_t seg0 = g0 > 0 ? g0 : 0;
    int64_t seg1 = (g0 + p.csz < gend) ? g0 + p.csz : gend;
    seg1 = seg1 < p.B ? seg1 : p.B;
    const int64_t b0 = seg0 + sub * p.spc;
    const int64_t b1 = (b0 + p.spc < seg1) ? b0 + p.spc : seg1;
    if (b0 >= b1) return;

    {
        const float* We = p.W + (int64_t)e * D;
        for (int i = tid; i < D; i += 256) flat[i] = We[i];
        if (tid == 0) flat[ZERO_IDX] = 0.0f;
    }
    __syncthreads();
    const float m1 = laundered_minus_one();
    // feature_nn weights -> bf16 parts in registers (every wave builds its own copy)
    Frag<NS> A1[3], A2[3][2], A3[2][2];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) A1[mt] = lowp_wfrag<NS, H>(flat, 0, mt, 0, g, c, m1);
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int s = 0; s < 2; ++s) A2[mt][s] = lowp_wfrag<NS, H>(flat, 1, mt, s, g, c, m1);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int s = 0; s < 2; ++s) A3[mt][s] = lowp_wfrag<NS, H>(flat, 2, mt, s, g, c, m1);
    {   // regress_nn operands (exact fp32) replace the flat vector in place
        constexpr int PER = (NF2 + 3) / 4;
        int idx[PER];
        float tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int f = wave + 4 * i;
            idx[i] = p.tab_f2[(f < NF2 ? f : NF2 - 1) * 64 + lane];
        }
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
    float* epsscr = scr + wave * SCRL;  // Philox normals of the finish; the x-tile staging buffer during the tile loop
    float* sumscr = epsscr + XST;
    float* xst = epsscr;
    static_assert(XST >= 8 * 16 * 4 && XST >= 16 * S2 && XST % 4 == 0, "staging buffer covers the tile and the Philox scratch, 16-byte aligned");
    // x tile staging (global -> registers -> LDS -> fragment registers).  A lane's fragment is 8 floats of ONE row, and lanes
    // that are neighbours in the wave hold DIFFERENT rows (164 B apart), so loading fragments straight from global memory makes
    // every lane touch its own cache line: 64 lines per wave-instruction, and the kernel ran at exactly one tile per ~190 cycles
    // per CU -- the L1's rate for that pattern, not the matrix or vector pipes'.  So a tile is fetched by ROW CHUNKS: the 16 rows
    // (4 systems x 4 timesteps) x 8 chunks of 16 B = columns 8..11, ..., 32..35 and [36, 37, column 0, 1.0] -- exactly the four
    // k-groups' fragments (chunks 2g, 2g + 1), group 3's "column 0" and "constant 1.0 of the bias slot" taking the places of the
    // dead columns 38, 39 -- two chunks per lane: round 0 chunks 0..3, round 1 chunks 4..7; four consecutive lanes read 64
    // consecutive bytes of a row.
    // LDS layout and banks (MI355X_MICROARCH.md, LDS): ds_read_b128 is served in four 16-lane groups over 64 banks (16 slots of
    // 16 B), each group holding every row c exactly once (with two different k-groups g); ds_write_b128 in eight groups of 8
    // contiguous lanes over 32 banks (8 slots).  Slot of (chunk k, row r) = 16 k + (r ^ (k & 3)):
    //   reads:  slot mod 16 = c ^ (k & 3): the XOR permutes rows inside aligned blocks of 4, so the 16 rows of a group stay
    //           16 different slots whatever chunks its lanes read -> conflict-free;
    //   writes: an 8-lane group stores rows {ra, ra + 4} x chunks {kb .. kb + 3}: slot mod 8 = ((ra & 3) ^ q) | 4 h, all different
    //           -> conflict-free.   (Round 2 staged rows verbatim at their 41-float stride: 27 % of the LDS cycles were conflicts.)
    const int sq4 = lane & 3, sh = (lane >> 2) & 1, su = lane >> 3;
    const int srow = (su & 3) + 8 * (su >> 2) + 4 * sh;   // this lane's staging row; its chunks: sq4 (round 0) and 4 + sq4 (round 1)
    for (int64_t wb0 = b0 + (int64_t)wave * 16; wb0 < b1; wb0 += 64) {
        for (int sb = 0; sb < 4; ++sb) {  // four tiles' worth of systems: wb0 + 4 sb + sq
            if (wb0 + 4 * sb >= b1) break;  // wave-uniform
            const int64_t sys = wb0 + 4 * sb + sq;
            const bool valid = sys < b1;
            const int64_t sysc = valid ? sys : b1 - 1;
            // staging chunk addresses = wave-uniform base of the tile (scalar registers, advanced on the scalar unit) + a per-lane
            // 32-bit offset that does not change over the tiles: no vector address arithmetic in the loop
            const float* tbase = p.x + (wb0 + 4 * sb) * rowstride;
            uint32_t soff, soff0;   // offset of this lane's chunk 0 of the row (column 8 + 4 sq4), offset of the row's column 0
            {
                int64_t ss = wb0 + 4 * sb + (srow >> 2);           // row = 4 * (system of the tile) + timestep phase
                ss = ss < b1 ? ss : b1 - 1;                         // >= wb0 + 4 sb: the offset is not negative
                soff0 = (uint32_t)((ss - (wb0 + 4 * sb)) * rowstride + (srow & 3) * F);
                soff = soff0 + 8 + 4 * sq4;
            }
            // this lane's fragment in the staging buffer: chunks 2g and 2g + 1 of its row c
            const f32x4* fr0 = reinterpret_cast<const f32x4*>(xst) + xslot(2 * g, c);
            const f32x4* fr1 = reinterpret_cast<const f32x4*>(xst) + xslot(2 * g + 1, c);
            f32x4* const st0 = reinterpret_cast<f32x4*>(xst) + xslot(sq4, srow);
            f32x4* const st1 = reinterpret_cast<f32x4*>(xst) + xslot(4 + sq4, srow);

            f32x4 mean0 = {0, 0, 0, 0}, m20 = {0, 0, 0, 0}, mean1 = {0, 0, 0, 0}, m21 = {0, 0, 0, 0};
            // Three tiles are kept in flight per wave (one tile = 2.6 KB; with ONE in flight the kernel ran at exactly
            // tile bytes / loaded memory latency (~2 200 cycles) per wave, whatever the arithmetic): three register slots P, Q, R
            // rotate by unrolling the tile loop three times.
            constexpr int DEPTH = NS == 3 ? 1 : 3;  // the six-product form has no registers to spare (and is bound by its arithmetic)
            struct Slot { f32x4 v[2]; float c0; };
            Slot s0, s1, s2;
            auto fetch = [&](int it, Slot& slot) {
                const int itc = it < ntiles ? it : ntiles - 1;
                const float* tb = tbase + (int64_t)itc * 4 * F;
                slot.v[0] = *reinterpret_cast<const f32x4u*>(tb + soff);        // columns 8 + 4 sq4 ..
                slot.v[1] = *reinterpret_cast<const f32x4u*>(tb + soff + 16);   // columns 24 + 4 sq4 .. (sq4 = 3: 36..39)
                slot.c0 = tb[soff0];
            };
            auto stage = [&](const Slot& slot) {
                f32x4 v = slot.v[1];
                v.z = sq4 == 3 ? slot.c0 : v.z;   // chunk 7 = [36, 37, column 0, 1.0]
                v.w = sq4 == 3 ? 1.0f : v.w;
                *st0 = slot.v[0];
                *st1 = v;
            };
            fetch(0, s0);
            if constexpr (DEPTH == 3) {
                fetch(1, s1);
                fetch(2, s2);
            }
            stage(s0);
            Frag<NS> Bs0, Bs1;  // B operands of layers 2 and 3; the constant half of Bs1 (bias slot, padding) is written once
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                Bs1.part[q][2] = q == 0 ? ONE_LO<H> : 0u;  // element 4 = 1.0, element 5 = 0
                Bs1.part[q][3] = 0u;
            }
            // one tile: LDS holds tile `it`; P holds tile it+1, R is free (gets tile it+3); at the end P goes to LDS
            auto tile = [&](int it, const Slot& P, Slot& R) {

                // layer-1 B operand from the staged tile: 8 floats of this lane's row at column 8 + 8g (group 3: 32..37, then
                // column 0 and the constant 1.0 of the bias slot, put there by stage())
                Frag<NS> B1;
                {
                    const f32x4 va = *fr0, vb = *fr1;     // two conflict-free ds_read_b128
                    const float v[6] = {va.x, va.y, va.z, va.w, vb.x, vb.y};
                    const float v6 = vb.z, v7 = vb.w;
                    uint32_t pk[NS];
                    split_pair<NS, H>(v[0], v[1], m1, pk);
#pragma unroll
                    for (int q = 0; q < NS; ++q) B1.part[q][0] = pk[q];
                    split_pair<NS, H>(v[2], v[3], m1, pk);
#pragma unroll
                    for (int q = 0; q < NS; ++q) B1.part[q][1] = pk[q];
                    split_pair<NS, H>(v[4], v[5], m1, pk);
#pragma unroll
                    for (int q = 0; q < NS; ++q) B1.part[q][2] = pk[q];
                    split_pair<NS, H>(v6, v7, m1, pk);
#pragma unroll
                    for (int q = 0; q < NS; ++q) B1.part[q][3] = pk[q];
                }
                fetch(it + DEPTH, R);  // DEPTH tiles ahead: in flight while this tile (and the next two) compute
                f32x4 acc[3];
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) acc[mt] = mfma_split<NS, H>(A1[mt], B1, (f32x4){0, 0, 0, 0});
                lowp_next_operand<NS, H>(acc, m1, Bs0, Bs1);
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    acc[mt] = mfma_split<NS, H>(A2[mt][0], Bs0, (f32x4){0, 0, 0, 0});
                    acc[mt] = mfma_split<NS, H>(A2[mt][1], Bs1, acc[mt]);
                }
                lowp_next_operand<NS, H>(acc, m1, Bs0, Bs1);
                f32x4 y0 = mfma_split<NS, H>(A3[0][0], Bs0, (f32x4){0, 0, 0, 0});
                y0 = mfma_split<NS, H>(A3[0][1], Bs1, y0);
                f32x4 y1 = mfma_split<NS, H>(A3[1][0], Bs0, (f32x4){0, 0, 0, 0});
                y1 = mfma_split<NS, H>(A3[1][1], Bs1, y1);
                // torch.mean / torch.std over time (:418-419).  y0[r] = neuron 4g + r, y1[r] = neuron 16 + 4g + r (a real neuron
                // only for g = 0).  NS >= 2: Welford over this lane's timesteps, as in the fp32 kernel.  NS = 1: plain sums of y and
                // y^2 (half the instructions; their fp32 cancellation error, ~1e-5 relative on the variance, is far below bf16's own).
                auto lo2 = [](const f32x4& v) { return (f32x2){v.x, v.y}; };
                auto hi2 = [](const f32x4& v) { return (f32x2){v.z, v.w}; };
                auto put = [](f32x4& v, f32x2 a, f32x2 b) { v = (f32x4){a.x, a.y, b.x, b.y}; };
                if constexpr (NS == 1) {   // two-element vector arithmetic: v_pk_add_f32 / v_pk_fma_f32 (no SLP vectoriser in this unit)
                    auto acc = [&](f32x4& mean, f32x4& m2, const f32x4& y) {
                        put(mean, lo2(mean) + lo2(y), hi2(mean) + hi2(y));
                        put(m2, __builtin_elementwise_fma(lo2(y), lo2(y), lo2(m2)), __builtin_elementwise_fma(hi2(y), hi2(y), hi2(m2)));
                    };
                    acc(mean0, m20, y0);
                    acc(mean1, m21, y1);
                } else {
                    const float rcn = p.rcp_tab[it];
                    const f32x2 rc2 = {rcn, rcn};
                    auto welford = [&](f32x4& mean, f32x4& m2, const f32x4& y) {
                        const f32x2 dl0 = lo2(y) - lo2(mean), dl1 = hi2(y) - hi2(mean);
                        const f32x2 mn0 = __builtin_elementwise_fma(dl0, rc2, lo2(mean)), mn1 = __builtin_elementwise_fma(dl1, rc2, hi2(mean));
                        put(m2, __builtin_elementwise_fma(dl0, lo2(y) - mn0, lo2(m2)), __builtin_elementwise_fma(dl1, hi2(y) - mn1, hi2(m2)));
                        put(mean, mn0, mn1);
                    };
                    welford(mean0, m20, y0);
                    welford(mean1, m21, y1);
                }
                // this tile's fragment reads were issued at its top: the buffer is free for the next tile.  The barrier keeps the
                // writes (and the wait for the loads they need) HERE, a whole tile of work after the loads were issued: left alone,
                // the scheduler hoists them to just behind the loads and every tile waits out a full memory latency.
                __builtin_amdgcn_sched_barrier(0);
                stage(P);
            };
            if constexpr (DEPTH == 3) {
                for (int it = 0; it < ntiles; it += 3) {
                    tile(it, s1, s0);
                    if (it + 1 < ntiles) tile(it + 1, s2, s1);
                    if (it + 2 < ntiles) tile(it + 2, s0, s2);
                }
            } else {
                for (int it = 0; it < ntiles; ++it) tile(it, s0, s0);
            }
            if constexpr (NS == 1) {  // sums over the 4 timestep phases, then mean and M2 = sum y^2 - (sum y)^2 / T
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a = mean0[i] + quad_perm<0xB1>(mean0[i]), b = m20[i] + quad_perm<0xB1>(m20[i]);
                    a = a + quad_perm<0x4E>(a); b = b + quad_perm<0x4E>(b);
                    mean0[i] = a / nT; m20[i] = fmaxf(b - a * mean0[i], 0.0f);  // cancellation can leave a tiny negative number
                    a = mean1[i] + quad_perm<0xB1>(mean1[i]); b = m21[i] + quad_perm<0xB1>(m21[i]);
                    a = a + quad_perm<0x4E>(a); b = b + quad_perm<0x4E>(b);
                    mean1[i] = a / nT; m21[i] = fmaxf(b - a * mean1[i], 0.0f);
                }
            } else {
            // merge the 4 timestep phases (lanes c, c^1, c^2, c^3): equal-count Chan update, symmetric
            {
                float half_n = half_n0;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float om = st == 0 ? quad_perm<0xB1>(mean0[i]) : quad_perm<0x4E>(mean0[i]);
                        float o2 = st == 0 ? quad_perm<0xB1>(m20[i]) : quad_perm<0x4E>(m20[i]);
                        float dl = om - mean0[i];
                        m20[i] = (m20[i] + o2) + (dl * dl) * half_n;
                        mean0[i] = (mean0[i] + om) * 0.5f;
                        om = st == 0 ? quad_perm<0xB1>(mean1[i]) : quad_perm<0x4E>(mean1[i]);
                        o2 = st == 0 ? quad_perm<0xB1>(m21[i]) : quad_perm<0x4E>(m21[i]);
                        dl = om - mean1[i];
                        m21[i] = (m21[i] + o2) + (dl * dl) * half_n;
                        mean1[i] = (mean1[i] + om) * 0.5f;
                    }
                    half_n = half_n * 2.0f;
                }
            }
            }
            // sampled moments (:420-431).  Lane (g, tph) finishes neuron 4g + tph and, for g = 0, neuron 16 + tph.
            const int slot = 4 * sb + sq;  // this system's slot among the 16 of the wave-batch
            if (p.eps == nullptr) {        // the system's 40 normals = ten Philox blocks, one per lane li < 10 of its 16 lanes
                const int li = 4 * g + tph;
                if (li < 10) *reinterpret_cast<f32x4*>(epsscr + slot * S2 + 4 * li) = philox_eps4(p.row_id0 + r, p.sys_id0 + sysc, li, p.seed);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                if (part == 1 && g != 0) continue;
                const f32x4 mv = part == 0 ? mean0 : mean1, qv = part == 0 ? m20 : m21;
                const int n = part == 0 ? 4 * g + tph : 16 + tph;
                const float sample_mu = tph == 0 ? mv[0] : tph == 1 ? mv[1] : tph == 2 ? mv[2] : mv[3];
                const float msum = tph == 0 ? qv[0] : tph == 1 ? qv[1] : tph == 2 ? qv[2] : qv[3];
                float e1, e2;
                if (p.eps) {
                    const float* ep = p.eps + (r * p.B + sysc) * S2;
                    e1 = ep[n]; e2 = ep[L + n];
                } else {
                    e1 = epsscr[slot * S2 + n]; e2 = epsscr[slot * S2 + L + n];
                }
                float mu_s, sd_s;
                sampled_moments(sample_mu, msum, e1, e2, nm1, nT, mu_s, sd_s);
                sumscr[slot * S2 + n] = mu_s;
                sumscr[slot * S2 + L + n] = sd_s;
                if (p.summary && valid) {
                    float* sp = p.summary + (r * p.B + sys) * S2 + n;
                    sp[0] = mu_s;
                    sp[L] = sd_s;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- regress_nn on the 16 systems of this wave-batch, exact fp32 (16x16x4 path): column c <-> system wb0 + c
        const int64_t sysb = wb0 + c;
        const bool validb = sysb < b1;
        float skeep[10];
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) skeep[ks] = sumscr[c * S2 + kmap_summary(ks, g)];
        const f32x4 a6 = regress16<false>(skeep, f2frag, lane);   // the one routine of the fp32 kernel (bnn_common.hip.h)
        if (g == 0 && validb) {
            const f32x2 ms = soft_clamp2(a6[0], a6[1], p.std_lo, p.std_span);
            const int64_t o = (r * p.B + sysb) * 2;
            *reinterpret_cast<f32x2*>(p.out + o) = ms;
            if (p.pre_clamp) *reinterpret_cast<f32x2*>(p.pre_clamp + o) = (f32x2){a6[0], a6[1]};
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NS, bool H>
inline hipError_t launch_lowp_form(unsigned nblk, hipStream_t st, const FwdParams& p) {
    allow_big_lds<&bnn_forward_lowp_kernel<NS, H>>();   // once per (function, device), thread-safe
    hipLaunchKernelGGL((bnn_forward_lowp_kernel<NS, H>), dim3(nblk), dim3(256), lowp_lds_bytes(), st, p);
    return hipGetLastError();
}

}  // namespace bnn
