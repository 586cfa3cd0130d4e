// bnn_generic.cpp -- host side of the generic forward engine: turns (n_features, hidden, latent, depth in / out, fix_megno) into
// the layer descriptor the kernel walks (bnn_generic.h), picks the register bucket and fits the LDS budget.
#include "bnn_generic.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>

namespace bnn {

namespace {
constexpr int LDS_BYTES = 160 * 1024;   // per CU on gfx950; one workgroup may take all of it
constexpr int HQ_BUCKETS[] = {12, 16, 24, 32};

int mlp_nlin(int layers) { return layers == 0 ? 1 : layers + 2; }   // mlp() (spock_reg_model.py:301-321)
}  // namespace

static int gen_build_impl(int F, int H, int L, int depth_in, int depth_out, bool megno, bool spec, int w8, uint64_t drop_mask, int pool_regs,
                          GenArch* out, const char** why) {
    static const char* msg_f = "n_features must be 41 or 82 (time_series_features x (1 + include_derivatives))";
    static const char* msg_w = "hidden and latent must be in [1, 128] and the summary width 2 latent (+ 2) at most 128";
    static const char* msg_d = "depth `in` / `out` must be >= 0 with at most 16 Linear modules in the two MLPs together";
    static const char* msg_8 = "that many waves' pool state does not fit LDS next to this network's weight image";
    static const char* msg_l = "feature_nn's weights do not fit the 160 KB of LDS (roughly F*H + in*H*H + H*L <= 36 000 floats)";
    if (F != 41 && F != 82) { *why = msg_f; return -2; }
    const int SM = 2 * L + (megno ? 2 : 0);
    if (H < 1 || H > GEN_MAX_WIDTH || L < 1 || L > GEN_MAX_WIDTH || SM > GEN_MAX_WIDTH) { *why = msg_w; return -2; }
    if (depth_in < 0 || depth_out < 0 || mlp_nlin(depth_in) + mlp_nlin(depth_out) > GEN_MAX_LAYERS) { *why = msg_d; return -2; }
    GenArch g{};
    g.F = F; g.H = H; g.L = L; g.SM = SM; g.megno = megno ? 1 : 0;
    g.pool_lds = (spec && pool_regs) ? 0 : 1;
    g.n_feat = mlp_nlin(depth_in); g.n_reg = mlp_nlin(depth_out);
    g.lq = (L + 3) / 4; g.smq = (SM + 3) / 4;
    g.fq = F == 41 ? 11 : 21;
    g.nin_blocks = (F + 5) / 6;
    g.off_inlv = 0; g.off_sumlv = F;
    int off = F + SM, bias0 = 0, need = std::max(g.smq, g.lq);
    const int nl = g.n_feat + g.n_reg;
    for (int l = 0; l < nl; ++l) {
        const bool feat = l < g.n_feat;
        const int ll = feat ? l : l - g.n_feat, nn = feat ? g.n_feat : g.n_reg;
        GenLayer& y = g.layer[l];
        y.K = ll == 0 ? (feat ? F : SM) : H;
        if (l == 0 && spec && drop_mask) {   // layer 0 over the unmasked columns only
            int nlive = 0;
            for (int c = 0; c < F; ++c) nlive += !(c < 64 && ((drop_mask >> c) & 1ull));
            if (nlive < 1) { *why = "every input column is masked"; return -2; }
            if (nlive < F) { g.in_live = nlive; y.K = nlive; }
        }
        y.N = ll == nn - 1 ? (feat ? L : 2) : H;
        y.relu = ll < nn - 1 ? 1 : 0;
        y.nkq = (y.K + 3) / 4;
        const int groups = (y.N + 3) / 4;
        y.nblk = (groups + 3) / 4;
        y.ng_last = groups - 4 * (y.nblk - 1);
        y.off_w = off; off += y.N * (l == 0 ? F : y.K);
        y.off_b = off; off += y.N;
        y.bias0 = bias0; bias0 += 16 * y.nblk;
        y.wreg0 = -1;
        need = std::max(need, 4 * y.nblk);
        if (l > 0) need = std::max(need, y.nkq);
    }
    g.d = off;
    g.nbias = bias0;
    g.hq = 0;
    if (spec) {
        g.hq = 4 * ((need + 3) / 4);   // exact: the widest layer's blocks
    } else {
        for (int b : HQ_BUCKETS)
            if (!g.hq && need <= b) g.hq = b;
    }
    if (!g.hq || g.hq > 32) { *why = msg_w; return -2; }
    // LDS: feature_nn's registers must be resident; regress_nn's are if that costs no waves
    int nfeat_regs = 0, nreg_regs = 0;
    for (int l = 0; l < nl; ++l) (l < g.n_feat ? nfeat_regs : nreg_regs) += g.layer[l].nblk * g.layer[l].nkq;
    auto waves_for = [&](int nwreg, int reg_in_lds) {
        GenArch t = g;
        t.nwreg = nwreg;
        t.reg_in_lds = reg_in_lds;
        for (int nw : {16, 8, 4, 2, 1}) {   // eight waves (two per SIMD) only where the kernel fits 256 registers: the narrowest bucket
            if (nw == 16 && !(spec && w8 == 2)) continue;   // sixteen (four per SIMD, 128 registers): specialised forms of small networks, on request
            if (nw != 16 && spec && w8 == 2) continue;
            if (nw == 8 && (spec ? w8 == 0 : false)) continue;
            if (nw == 8 && (spec ? w8 < 0 : true) && (t.hq > GEN_W8_HQ || t.fq != 11)) continue;
            if (nw != 8 && spec && w8 == 1) continue;
            if ((int64_t)(gen_shared_floats(t) + nw * gen_wave_floats(t)) * 4 <= LDS_BYTES) return nw;
        }
        return 0;
    };
    const int w_all = waves_for(nfeat_regs + nreg_regs, 1), w_feat = waves_for(nfeat_regs, 0);
    if (!w_feat) { *why = (spec && w8 >= 1) ? msg_8 : msg_l; return -2; }
    g.reg_in_lds = (w_all >= w_feat) ? 1 : 0;
    g.nwaves = g.reg_in_lds ? w_all : w_feat;
    g.nwreg = nfeat_regs + (g.reg_in_lds ? nreg_regs : 0);
    int wreg = 0;
    for (int l = 0; l < nl; ++l)
        if (l < g.n_feat || g.reg_in_lds) { g.layer[l].wreg0 = wreg; wreg += g.layer[l].nblk * g.layer[l].nkq; }
    g.lds_bytes = (gen_shared_floats(g) + g.nwaves * gen_wave_floats(g)) * 4;
    *out = g;
    return 0;
}

int gen_build(int F, int H, int L, int depth_in, int depth_out, bool megno, GenArch* out, const char** why) {
    return gen_build_impl(F, H, L, depth_in, depth_out, megno, false, -1, 0, 0, out, why);
}

int gen_build_spec(int F, int H, int L, int depth_in, int depth_out, bool megno, int w8, uint64_t drop_mask, int pool_regs, GenArch* out,
                   const char** why) {
    return gen_build_impl(F, H, L, depth_in, depth_out, megno, true, w8, drop_mask, pool_regs, out, why);
}

// One form's policy struct + kernel.  tag = "" for a stand-alone (run-time compiled) file, else the suffix of the names in the embedded unit.
static void spec_form(std::string& s, const GenArch& g, int noisy, int pool_regs, int block_major, uint64_t drop_mask, const char* tag, int resident = 0) {
    char t[256];
    auto add = [&](const char* fmt, auto... a) { snprintf(t, sizeof t, fmt, a...); s += t; };
    add("namespace bnn {\nstruct SpecArch%s {\n", tag);
    add("    static constexpr bool kq_major = %s;\n", block_major ? "false" : "true");
    add("    static constexpr int n_feat = %d, n_reg = %d;\n", g.n_feat, g.n_reg);
    if (g.nwaves == 16) s += "    static constexpr bool x_late = true;   // four waves per SIMD at 128 registers: no row prefetch\n";
    if (resident && !block_major) {   // feature_nn's weight registers stay in VGPRs: the layer tables as constexpr functions
        int nw = 0;
        for (int l = 0; l < g.n_feat; ++l) nw += g.layer[l].nkq * g.layer[l].nblk;
        add("    static constexpr int n_wres = %d;   // feature_nn's weight registers, resident across the tiles\n", nw);
        auto table = [&](const char* name, auto field) {
            add("    static constexpr int %s(int l) {\n        constexpr int t[] = {", name);
            for (int l = 0; l < g.n_feat; ++l) add("%d, ", field(g.layer[l]));
            s += "};\n        return t[l];\n    }\n";
        };
        table("nkq", [](const GenLayer& y) { return y.nkq; });
        table("nblk", [](const GenLayer& y) { return y.nblk; });
        table("ng_last", [](const GenLayer& y) { return y.ng_last; });
        table("wreg0", [](const GenLayer& y) { return y.wreg0; });
        table("bias0", [](const GenLayer& y) { return y.bias0; });
        table("relu", [](const GenLayer& y) { return y.relu; });
    }
    add("    static constexpr int pool_lq = %d, lat_nfull = %d;   // Welford state of the pool in registers (0: in LDS)\n", pool_regs ? g.lq : 0,
        4 * (g.layer[g.n_feat - 1].nblk - 1));
    if (g.in_live) {   // layer 0 over the unmasked columns: logical input k -> column live(k) (padding slots repeat column live(0): zero weights)
        add("    static constexpr int in_q = %d;   // input quads of layer 0 after dropping the masked columns\n", g.layer[0].nkq);
        s += "    static constexpr int live(int k) {\n        constexpr int t[] = {";
        int first = -1, n = 0;
        for (int c = 0; c < g.F; ++c)
            if (!(c < 64 && ((drop_mask >> c) & 1ull))) {
                if (first < 0) first = c;
                add("%d, ", c);
                ++n;
            }
        for (; n < 4 * g.layer[0].nkq; ++n) add("%d, ", first);
        s += "};\n        return t[k];\n    }\n";
        unsigned xq = 0;   // quads of a row with a live column (or the raw MEGNO column of fix_megno, summarize_megno :480-484)
        for (int c = 0; c < g.F; ++c)
            if (!(c < 64 && ((drop_mask >> c) & 1ull)) || (g.megno && c == MEGNO_COL)) xq |= 1u << (c >> 2);
        add("    static constexpr uint32_t x_quads = 0x%xu;   // quads of an input row that are read at all\n", xq);
    }
    s += "    static DEVINL GenArch get(const GenParams&) {\n        constexpr GenArch value = {\n";
    add("            %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d,\n            {\n", g.F, g.H, g.L, g.SM, g.d, g.megno, g.n_feat,
        g.n_reg, g.nwreg, g.nbias, g.fq, g.hq, g.lq, g.smq, g.nin_blocks, g.reg_in_lds, g.nwaves, g.lds_bytes, g.off_inlv, g.off_sumlv, g.pool_lds, g.in_live);
    for (int l = 0; l < g.n_feat + g.n_reg; ++l) {
        const GenLayer& y = g.layer[l];
        add("                {%d, %d, %d, %d, %d, %d, %d, %d, %d, %d},\n", y.K, y.N, y.nkq, y.nblk, y.ng_last, y.off_w, y.off_b, y.wreg0, y.bias0, y.relu);
    }
    s += "            }};\n        return value;\n    }\n};\n}  // namespace bnn\n\n";
    const bool w8 = g.nwaves >= 8;
    add("extern \"C\" __global__ __launch_bounds__(%d, 1) void bnn_spec_forward%s(const bnn::GenParams P) {\n", g.nwaves == 16 ? 1024 : w8 ? 512 : 256, tag);
    add("    __shared__ __attribute__((aligned(16))) float lds[%d];\n", g.lds_bytes / 4);
    add("    bnn::generic_body<%d, %d, %s, bnn::SpecArch%s, %d>(P, lds);\n}\n", g.fq, g.hq, w8 ? "true" : "false", tag, noisy ? 1 : 0);
}

static int copy_out(const std::string& s, char* buf, size_t cap) {
    if (buf && cap) {
        const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int)s.size();
}

int gen_spec_source(const GenArch& g, int noisy, int pool_regs, int block_major, uint64_t drop_mask, char* buf, size_t cap, int resident) {
    std::string s = "// generated by bnn_spec_source (bnn_generic.cpp): the generic forward engine compiled for ONE network -- every shape a constant\n"
                    "#include \"bnn_generic.hip.h\"\n\n";
    spec_form(s, g, noisy, pool_regs, block_major, drop_mask, "", resident);
    return copy_out(s, buf, cap);
}

int gen_spec_embedded_source(char* buf, size_t cap) {
    GenArch q, n;
    const char* why = "";
    if (gen_build_spec(F, H, L, 1, 1, false, 1, V50_ZERO_MASK, 1, &q, &why) || gen_build_spec(F, H, L, 1, 1, false, 1, 0, 1, &n, &why)) return -2;
    std::string s = "// bnn_fwd_v50spec.hip -- GENERATED by bnn_spec_embedded_source (bnn_generic.cpp; scripts/regen_embedded.py), do not edit:\n"
                    "// the pretrained network's two specialised forms of the generic forward engine (DESIGN.md section 4.10), compiled into the\n"
                    "// library so that its ragged series lengths (T % 4 != 0, T < 8) need no compiler at run time.  Quiet form: the pretrained\n"
                    "// column mask; noisy form: any mask.  tests/test_spec_cpu.py regenerates this text and compares.\n"
                    "#include \"bnn_generic.hip.h\"\n\n";
    spec_form(s, q, 0, 1, 0, V50_ZERO_MASK, "_v50q");
    s += "\n";
    spec_form(s, n, 1, 1, 0, 0, "_v50n");
    s += "\nnamespace bnn {\nhipError_t launch_fwd_v50spec(bool noisy, unsigned nblk, hipStream_t st, const GenParams& P) {\n"
         "    if (noisy) hipLaunchKernelGGL(bnn_spec_forward_v50n, dim3(nblk), dim3(512), 0, st, P);\n"
         "    else hipLaunchKernelGGL(bnn_spec_forward_v50q, dim3(nblk), dim3(512), 0, st, P);\n"
         "    return hipGetLastError();\n}\n}  // namespace bnn\n";
    return copy_out(s, buf, cap);
}

}  // namespace bnn
