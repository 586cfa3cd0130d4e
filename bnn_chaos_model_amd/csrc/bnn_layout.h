// bnn_layout.h -- operand layout shared by the table builder (host) and the kernels (device).
//
// regress_nn is evaluated with v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain), 16 systems at a time.
// For every Linear layer  out[neuron][row] = sum_k W[neuron][k] * act[k][row] + b[neuron]:
//   A operand = weights      lane (g = lane>>4, m = lane&15) holds W[nmap(mt, m)][kmap(kstep, g)]
//   B operand = activations  lane (g, c = lane&15)           holds act[kmap(kstep, g)][row c]
//   C/D                      lane (g, c), register i         holds out[nmap(mt, 4g+i)][row c]
// so a layer's output registers ARE the next layer's B operands: k-step (mt', i') of the next
// layer consumes register i' of m-tile mt', whose lane group g carries neuron nmap(mt', 4g+i').
// The neuron->(m-tile, row) maps below put padding in whole registers so that a 40-wide layer
// costs 10 k-steps (not 12) downstream.
// feature_nn is evaluated with v_mfma_f32_4x4x1_16b_f32 from packed LDS images ("second operand layout" below).
#pragma once
#include <stdint.h>

#ifndef BNN_HD
#ifdef __HIPCC__
#define BNN_HD __host__ __device__
#else
#define BNN_HD
#endif
#endif

namespace bnn {

constexpr int F = 41;  // input features per timestep
constexpr int H = 40;  // hidden width
constexpr int L = 20;  // latent width (feature_nn output)
constexpr int S2 = 2 * L;

// flat parameter vector offsets (reference state_dict order, spock_reg_model.py:734-761)
constexpr int OFF_INLV = 0;
constexpr int OFF_SUMLV = OFF_INLV + F;       // 41
constexpr int OFF_W1 = OFF_SUMLV + S2;        // 81
constexpr int OFF_B1 = OFF_W1 + H * F;        // 1721
constexpr int OFF_W2 = OFF_B1 + H;            // 1761
constexpr int OFF_B2 = OFF_W2 + H * H;        // 3361
constexpr int OFF_W3 = OFF_B2 + H;            // 3401
constexpr int OFF_B3 = OFF_W3 + L * H;        // 4201
constexpr int OFF_W4 = OFF_B3 + L;            // 4221
constexpr int OFF_B4 = OFF_W4 + H * S2;       // 5821
constexpr int OFF_W5 = OFF_B4 + H;            // 5861
constexpr int OFF_B5 = OFF_W5 + H * H;        // 7461
constexpr int OFF_W6 = OFF_B5 + H;            // 7501
constexpr int OFF_B6 = OFF_W6 + 2 * H;        // 7581
constexpr int D = OFF_B6 + 2;                 // 7583
constexpr int ZERO_IDX = D;                   // LDS slot that always holds 0.0f
constexpr int FLAT_LDS = 7680;                // floats reserved for the flat vector (+zero slot, 16B multiple; fix_megno: 7665 + 1)

// The same layout for either value of hparams['fix_megno'] (spock_reg_model.py:360-362): with it the summary is SM = 42 wide
// ([mu_sample(20) | std_sample(20) | megno mean, megno std]), so summary_noise_logvar has 42 entries and regress_nn.0 is 40 x 42;
// everything behind summary_noise_logvar moves by 2, everything behind regress_nn.0.weight by 82 (d = 7665).
template <bool MEGNO>
struct Lay {
    static constexpr int SM = S2 + (MEGNO ? 2 : 0);
    static constexpr int INLV = 0, SUMLV = F;
    static constexpr int W1 = SUMLV + SM, B1 = W1 + H * F, W2 = B1 + H, B2 = W2 + H * H, W3 = B2 + H, B3 = W3 + L * H;
    static constexpr int W4 = B3 + L, B4 = W4 + H * SM, W5 = B4 + H, B5 = W5 + H * H, W6 = B5 + H, B6 = W6 + 2 * H;
    static constexpr int D = B6 + 2, ZERO = D;
    // regress_nn fragments (bnn_tables.cpp): k-steps of regress_nn.0 (the 11th carries the two MEGNO statistics in lane groups 0, 1)
    static constexpr int NK4 = MEGNO ? 11 : 10;
    static constexpr int F_L4 = 0, F_L5 = 3 * NK4, F_L6 = F_L5 + 30, F_B4 = F_L6 + 10, F_B5 = F_B4 + 12, F_B6 = F_B5 + 12, NF2 = F_B6 + 4;
};
static_assert(Lay<false>::D == D && Lay<false>::W4 == OFF_W4 && Lay<false>::B6 == OFF_B6, "the fixed constants above are Lay<false>");
static_assert(Lay<true>::D == 7665 && Lay<true>::D < FLAT_LDS && Lay<true>::NF2 * 64 <= FLAT_LDS, "fix_megno layout fits the LDS budget");
constexpr int MEGNO_COL = 7;                  // self.megno_location (:371)

// runtime mirror for the host-side table builder
struct LayoutRT {
    int SM, W1, B1, W2, B2, W3, B3, W4, B4, W5, B5, W6, B6, D, NK4, NF2;
};
template <bool MEGNO>
constexpr LayoutRT layout_rt() {
    using Y = Lay<MEGNO>;
    return LayoutRT{Y::SM, Y::W1, Y::B1, Y::W2, Y::B2, Y::W3, Y::B3, Y::W4, Y::B4, Y::W5, Y::B5, Y::W6, Y::B6, Y::D, Y::NK4, Y::NF2};
}
inline LayoutRT layout_of(bool megno) { return megno ? layout_rt<true>() : layout_rt<false>(); }

constexpr int MAXK = 32;  // SWAG rank supported (reference uses K = 30, run_swag.py:37)

// v50 mask: live columns {0, 8..37}
constexpr uint64_t V50_ZERO_MASK = (1ull << 7) | (1ull << 3) | (1ull << 6) | (1ull << 38) | (1ull << 39) | (1ull << 40) |
                                   (1ull << 1) | (1ull << 2) | (1ull << 4) | (1ull << 5);

// 40-wide layers: three m-tiles; neurons 32..39 sit in registers i = 0,1 of m-tile 2.
BNN_HD inline int nmap_hidden(int mt, int m) {
    if (mt < 2) return 16 * mt + m;
    int g = m >> 2, i = m & 3;
    return i < 2 ? 32 + 2 * g + i : -1;
}
// 2-wide output layer: one m-tile, rows 0 and 1.
BNN_HD inline int nmap_out(int m) { return m < 2 ? m : -1; }

// k-steps over a 40-wide activation: 10 steps (mt', i'): (0,0..3), (1,0..3), (2,0..1).
constexpr int NKH = 10;
BNN_HD inline int kmap_hidden(int ks, int g) {
    int mt = ks >> 2, i = ks & 3;
    return nmap_hidden(mt, 4 * g + i);
}
// k-steps over the 40-wide summary [mu_sample(20) | std_sample(20)]: ks = kind*5 + r; lane group g carries neuron
// 4g + r for r < 4 and neuron 16 + g for r = 4 of that kind.
BNN_HD inline int kmap_summary(int ks, int g) {
    if (ks >= 10) return g < 2 ? S2 + g : -1;   // fix_megno: 11th k-step = [megno mean, megno std, 0, 0]
    int kind = ks / 5, r = ks % 5;
    int n = r < 4 ? 4 * g + r : 16 + g;
    return kind * L + n;
}
// ---- second operand layout: v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 neurons x 4 rows, K = 1) -----------------
// lane l = row (B operand), register r of neuron group n = neuron 4n + r.  The MFMA is issued with CBSZ = 4, ABID = a: the A
// operand (4 neurons x 1 input) held by lanes 4a .. 4a+3 is broadcast to all 16 blocks.  So ONE weight register carries the A
// operands of 16 MFMAs and all of feature_nn lives in registers for the workgroup's lifetime:
//   layer with G neuron groups and K inputs: MFMA number m = k * G + n  (input k, group n; per output: bias, then k ascending)
//   weight register R = m >> 4 of the layer, lane 4a + i (a = m & 15)  ->  W[neuron 4n + i][col(k)]
// No padding inside a layer: 40 outputs = 10 groups, 20 outputs = 5 groups; only the last register of a layer may have unused
// lanes (they read the zero slot).  Biases sit in a small LDS image [b1 | b2 | b3] and enter as the C operand of the first MFMA
// of a chain.
// KIN = number of layer-1 inputs the kernel multiplies: 31 for the v50 mask (live columns 0, 8..37, ascending), else all 41
// columns (weights of masked columns are zero in the registers).
BNN_HD inline int col4(int kin, int k) { return kin == F ? k : (k == 0 ? 0 : 7 + k); }
template <int KIN>
struct WR {
    static constexpr int G1 = H / 4, G2 = H / 4, G3 = L / 4;   // neuron groups per layer
    static constexpr int M1 = KIN * G1, M2 = H * G2, M3 = H * G3;  // MFMAs per 64-row tile: 310 (410) + 400 + 200
    static constexpr int R1 = (M1 + 15) / 16, R2 = (M2 + 15) / 16, R3 = (M3 + 15) / 16;
    static constexpr int NR = R1 + R2 + R3;                    // 58 (KIN = 31) / 64 (KIN = 41) weight registers
};
constexpr int BIAS_PAD = 112;  // LDS floats of the bias image (2 * H + L = 100, padded)

// regress_nn fragment count (see bnn_tables.cpp) for fix_megno = False (Lay<MEGNO>::NF2 in general)
constexpr int NF2 = 30 + 30 + 10 + 12 + 12 + 4;
static_assert(NF2 == Lay<false>::NF2, "");

}  // namespace bnn
