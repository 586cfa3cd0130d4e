// bnn_generic.h -- descriptor of the GENERIC forward engine: the network the reference builds from hparams
// (spock_reg_model.py:301-321 mlp(), :346-362: any `hidden`, `latent`, depth `in` / `out`, 41 or 82 features, fix_megno), any
// series length T >= 2.  Shared by the host (plan construction, bnn_abi.hip) and the kernel (bnn_generic.hip.h).
//
// Engine: lane = row as in the v50 kernel, v_mfma_f32_4x4x1_16b_f32 with the CBSZ/ABID broadcast, but the
// weight registers are STREAMED from an LDS image instead of living in VGPRs: one ds_read_b32 (256 B per wave) feeds 16 MFMAs =
// 4 consecutive inputs x 4 neuron groups.  A Linear layer with K inputs and N outputs is cut into
//   nblk = ceil(N / 16) output blocks of 4 neuron groups (16 neurons; the last block may hold 1..4 groups: ng_last),
//   nkq  = ceil(K / 4) input quads,
// and weight register (nb, kq) of the layer -- LDS image entry [(wreg0 + nb * nkq + kq) * 64 + lane] -- holds, in lane 4a + i with
// a = 4 kk + q:  W[neuron 16 nb + 4 q + i][input 4 kq + kk]   (zero outside the layer or on a masked input column).
// MFMA a of the register multiplies input 4 kq + kk into neuron group 4 nb + q; per output the accumulation order is bias, then
// the inputs ascending: the oracle's natural order, and bit for bit the v50 kernel's for feature_nn.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "bnn_layout.h"

namespace bnn {

constexpr int GEN_W8_HQ = 16;   // widest activation bucket (41-feature forms) that also has an eight-wave form, compiled for 256 registers
constexpr int GEN_MAX_LAYERS = 16;   // Linear modules of feature_nn and regress_nn together
constexpr int GEN_MAX_WIDTH = 128;   // hidden, latent and summary width (2 latent + 2) supported by the register buckets

struct GenLayer {
    int32_t K, N;         // inputs, outputs of the Linear module
    int32_t nkq, nblk;    // input quads, output blocks of 16 neurons
    int32_t ng_last;      // neuron groups (of 4) in the last block: 1..4
    int32_t off_w, off_b; // offsets of weight [N,K] and bias [N] in the flat parameter vector
    int32_t wreg0;        // first weight register of the layer in the LDS image (regress_nn layers kept out of LDS: -1)
    int32_t bias0;        // first float of the layer's bias image (16 * nblk floats, zero padded)
    int32_t relu;         // nn.ReLU behind it (every Linear of an mlp() but the last)
};

// Chan merge of two Welford partitions with counts (na, nb) (oracle/bnn_oracle.c merge_consts): mode 0 = equal counts, symmetric
// form, w1 = na / 2; 1 = general form, w1 = nb / n, w2 = na nb / n (float64 quotients rounded once); 2 = keep a; 3 = keep b.
struct GenMerge {
    int32_t mode;
    float w1, w2;
};

struct GenArch {
    int32_t F, H, L, SM, d;       // features, hidden, latent, summary width (2L, +2 with fix_megno), parameter count
    int32_t megno;
    int32_t n_feat, n_reg;        // Linear modules of feature_nn / regress_nn: layer[0 .. n_feat) / layer[n_feat .. n_feat + n_reg)
    int32_t nwreg, nbias;         // LDS image sizes: (nwreg + 1) * 64 floats of weight registers, nbias floats of biases
    int32_t fq, hq;               // register buckets the kernel was instantiated for: input quads (11 | 21), activation quads
    int32_t lq, smq;              // latent groups ceil(L / 4), summary quads ceil(SM / 4)
    int32_t nin_blocks;           // Philox blocks of six normals per row of input noise: ceil(F / 6)
    int32_t reg_in_lds;           // 1: regress_nn's weight registers are in the LDS image too; 0: gathered from the flat vector
    int32_t nwaves;               // waves per workgroup (8 for the narrowest bucket, else 4; 2 or 1 when the LDS budget forces it)
    int32_t lds_bytes;            // dynamic LDS of the launch
    int32_t off_inlv, off_sumlv;  // input_noise_logvar [F], summary_noise_logvar [SM]
    int32_t pool_lds;             // 1: the pool's Welford state has LDS rows (2 lq KB per wave); 0: it lives in registers and the four partitions
                                  // are merged by DPP (specialised forms with BNN_SPEC_POOL_REGS): no pool rows at all
    int32_t in_live;              // specialised quiet forms: layer 0 multiplies only the in_live unmasked columns (layer[0].K = in_live, its
                                  // weight rows are still F apart); 0 = every column (masked ones as zero weights)
    GenLayer layer[GEN_MAX_LAYERS];
};

// Host: build the descriptor.  Returns 0, or a negative code with *why set (unsupported width / depth, LDS budget).
//   depth_in / depth_out = hparams['in'] / hparams['out'] (the `layers` argument of mlp()).
int gen_build(int n_features, int hidden, int latent, int depth_in, int depth_out, bool megno, GenArch* out, const char** why);

// Host: the descriptor of the network's SPECIALISED form (compiled at run time for this one network, bnn_spec_source): the activation
// arrays are sized exactly (hq = 4 x the widest layer's blocks) instead of by bucket; w8 = 1 asks for the eight-wave / 256-register
// form (refused when eight waves' LDS does not fit), 0 for four waves (or fewer) at 512 registers, -1 lets the builder choose.
// drop_mask: input columns (bits below 64) layer 0 leaves out altogether -- the plan's zero mask for the quiet form (the same sums: a
// masked column only ever added +0), 0 for the noisy form (masked columns carry noise there) and for the block-major variant.
// pool_regs: the BNN_SPEC_POOL_REGS variant (no pool rows in LDS: more waves fit next to a large image).
int gen_build_spec(int n_features, int hidden, int latent, int depth_in, int depth_out, bool megno, int w8, uint64_t drop_mask, int pool_regs,
                   GenArch* out, const char** why);

// Host: the HIP source of that form -- `static constexpr GenArch` + one extern "C" kernel `bnn_spec_forward` around generic_body.
// Returns the length of the text (without the terminator); writes at most cap bytes.
int gen_spec_source(const GenArch& g, int noisy, int pool_regs, int block_major, uint64_t drop_mask, char* buf, size_t cap, int resident = 0);
// Host: the text of bnn_fwd_v50spec.hip -- the pretrained network's two specialised forms (eight waves, pool in registers; quiet under the
// pretrained mask, noisy under any) as one translation unit of the library, with their launcher.
int gen_spec_embedded_source(char* buf, size_t cap);

// Per-wave LDS floats: the pool state (mean, M2 per latent group, lane-major) while the tiles run and -- IN THE SAME ROWS, once the
// partitions are merged -- the staging area for one block of regress_nn's weight registers when those are not in the image
// (gen_pool_stage_floats = the larger of the two); then the summaries / Philox scratch and the MEGNO partitions.
BNN_HD inline int gen_sum_stride(const GenArch& g) { return 4 * g.smq; }   // (the 2 L pool normals share the summaries' rows: 2 L <= SM)
BNN_HD inline int gen_pool_stage_floats(const GenArch& g) {
    const int pool = g.pool_lds ? 2 * g.lq * 256 : 0, stage = g.reg_in_lds ? 0 : g.hq * 64;
    return pool > stage ? pool : stage;
}
BNN_HD inline int gen_wave_floats(const GenArch& g) { return gen_pool_stage_floats(g) + 16 * gen_sum_stride(g) + 128; }
// Workgroup-shared LDS floats: weight registers (+ one pad register for the read-ahead), biases, noise scales.
BNN_HD inline int gen_wimg_floats(const GenArch& g) { return (g.nwreg + 1) * 64; }
BNN_HD inline int gen_nsc_floats(const GenArch& g) { return 4 * ((g.F + 3) / 4) + 4 * ((g.SM + 3) / 4) + 8 * g.nin_blocks * 2; }
BNN_HD inline int gen_shared_floats(const GenArch& g) { return gen_wimg_floats(g) + g.nbias + gen_nsc_floats(g); }

}  // namespace bnn
