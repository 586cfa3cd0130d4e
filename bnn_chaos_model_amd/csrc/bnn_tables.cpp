// bnn_tables.cpp -- builds the gather tables that turn the reference's flat parameter vector (spock_reg_model.py:734-761
// order) into MFMA operands: the register-resident A operands of feature_nn (v_mfma_f32_4x4x1 with CBSZ broadcast) and the per-lane
// fragments + C-init biases of regress_nn (v_mfma_f32_16x16x4).
#include "bnn_tables.h"

namespace bnn {

namespace {

struct Builder {
    std::vector<int16_t>& t;
    int f = 0;
    explicit Builder(std::vector<int16_t>& tab) : t(tab) {}
    template <class Fn>
    void frag(Fn idx_of_lane) {
        for (int lane = 0; lane < 64; ++lane) t[(size_t)f * 64 + lane] = (int16_t)idx_of_lane(lane >> 4, lane & 15);
        ++f;
    }
};

}  // namespace

Tables build_tables(uint64_t zero_mask, bool all_columns, bool megno) {
    Tables T;
    const LayoutRT Y = layout_of(megno);
    // the names below shadow the fix_megno = False constants of bnn_layout.h on purpose: every index in this function is layout-relative
    const int OFF_W1 = Y.W1, OFF_B1 = Y.B1, OFF_W2 = Y.W2, OFF_B2 = Y.B2, OFF_W3 = Y.W3, OFF_B3 = Y.B3, OFF_W4 = Y.W4, OFF_B4 = Y.B4,
              OFF_W5 = Y.W5, OFF_B5 = Y.B5, OFF_W6 = Y.W6, OFF_B6 = Y.B6, ZERO_IDX = Y.D;
    (void)OFF_B1; (void)OFF_B2; (void)OFF_B3;
    // the v50 column mask (31 live columns) whichever of fix_megno / fix_megno2 zeroes column 7
    const bool v50 = !all_columns && zero_mask == V50_ZERO_MASK;
    auto dropped = [&](int col) { return !all_columns && ((zero_mask >> col) & 1ull); };

    T.f2.assign((size_t)Y.NF2 * 64, (int16_t)ZERO_IDX);
    Builder b2(T.f2);
    // layer 4 (regress_nn.0): k over the summary vector (fix_megno: an 11th k-step with the two MEGNO statistics)
    for (int ks = 0; ks < Y.NK4; ++ks)
        for (int mt = 0; mt < 3; ++mt)
            b2.frag([&](int g, int m) {
                int n = nmap_hidden(mt, m), k = kmap_summary(ks, g);
                return (n < 0 || k < 0) ? ZERO_IDX : OFF_W4 + n * Y.SM + k;
            });
    // layer 5 (regress_nn.2)
    for (int ks = 0; ks < NKH; ++ks)
        for (int mt = 0; mt < 3; ++mt)
            b2.frag([&](int g, int m) {
                int n = nmap_hidden(mt, m), k = kmap_hidden(ks, g);
                return (n < 0 || k < 0) ? ZERO_IDX : OFF_W5 + n * H + k;
            });
    // layer 6 (regress_nn.4)
    for (int ks = 0; ks < NKH; ++ks)
        b2.frag([&](int g, int m) {
            int n = nmap_out(m), k = kmap_hidden(ks, g);
            return (n < 0 || k < 0) ? ZERO_IDX : OFF_W6 + n * H + k;
        });
    for (int mt = 0; mt < 3; ++mt)
        for (int i = 0; i < 4; ++i)
            b2.frag([&](int g, int) {
                int n = nmap_hidden(mt, 4 * g + i);
                return n < 0 ? ZERO_IDX : OFF_B4 + n;
            });
    for (int mt = 0; mt < 3; ++mt)
        for (int i = 0; i < 4; ++i)
            b2.frag([&](int g, int) {
                int n = nmap_hidden(mt, 4 * g + i);
                return n < 0 ? ZERO_IDX : OFF_B5 + n;
            });
    for (int i = 0; i < 4; ++i)
        b2.frag([&](int g, int) {
            int n = nmap_out(4 * g + i);
            return n < 0 ? ZERO_IDX : OFF_B6 + n;
        });

    // 4x4x1 weight registers (bnn_layout.h, WR<KIN>): 31 live columns for the v50 mask, else all 41 with zero weights on the
    // masked ones.  Entry [R * 64 + lane]: lane 4a + i of register R holds W[neuron 4n + i][col(k)] of MFMA m = 16 R' + a = k * G + n
    // (R' = register index within the layer).
    T.kin4 = v50 ? 31 : F;
    {
        auto build4 = [&](auto lay) {
            using LY = decltype(lay);
            const int kin = T.kin4;
            T.f4.assign((size_t)LY::NR * 64, (int16_t)ZERO_IDX);
            auto layer = [&](int reg0, int nregs, int K, int G, int off_w, int ld, bool input) {
                for (int R = 0; R < nregs; ++R)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int m = 16 * R + (lane >> 2), i = lane & 3;
                        if (m >= K * G) continue;
                        const int k = m / G, n = m % G;
                        const int col = input ? col4(kin, k) : k;
                        if (input && dropped(col)) continue;
                        T.f4[(size_t)(reg0 + R) * 64 + lane] = (int16_t)(off_w + (4 * n + i) * ld + col);
                    }
            };
            layer(0, LY::R1, kin, LY::G1, OFF_W1, F, true);
            layer(LY::R1, LY::R2, H, LY::G2, OFF_W2, H, false);
            layer(LY::R1 + LY::R2, LY::R3, H, LY::G3, OFF_W3, H, false);
        };
        if (v50) build4(WR<31>{}); else build4(WR<F>{});
    }

    // accumulation order.  feature_nn (4x4x1, K = 1 per instruction): the accumulator starts at the bias, then the live inputs in
    // ascending order.  regress_nn (16x16x4): k-step major, lane group (the MFMA's k index) minor.
    for (int k = 0; k < F; ++k)
        if (!dropped(k)) T.order[0].push_back(k);
    for (int k = 0; k < H; ++k) { T.order[1].push_back(k); T.order[2].push_back(k); }
    for (int ks = 0; ks < NKH; ++ks)
        for (int g = 0; g < 4; ++g) {
            int k = kmap_hidden(ks, g);
            if (k < 0) continue;
            T.order[4].push_back(k);
            T.order[5].push_back(k);
        }
    for (int ks = 0; ks < Y.NK4; ++ks)
        for (int g = 0; g < 4; ++g)
            if (kmap_summary(ks, g) >= 0) T.order[3].push_back(kmap_summary(ks, g));
    return T;
}

}  // namespace bnn
