// bnn_abi_common.h -- what the translation units that implement the C ABI (include/bnn_chaos_hip.h) share on the HOST side: the
// per-thread error message, the plan, and the two helpers more than one unit needs.
//
//   bnn_abi.hip           plans, specialised forms, the forward entry points (forward / multiswag / statistics tail / slab drivers / latents /
//                         reduced precision) and the non-finite scan's entry point
//   bnn_ops_draw.hip      SWAGModel.sample_weights for J draws; the Philox fills
//   bnn_ops_reduce.hip    predictive moments; predict_instability on an explicit summary
//   bnn_ops_stats.hip     the evaluation scripts' post-sampling statistics: replay kernels, Philox epilogue, quantile sketch
//   bnn_ops_features.hip  feature packing + standardisation
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/bnn_chaos_hip.h"
#include "bnn_internal.h"
#include "bnn_tables.h"

// Sets the calling thread's bnn_last_error() message and returns `code` (defined in bnn_abi.hip).
int bnn_fail(int code, const std::string& msg);
static inline int fail(int code, const std::string& msg) { return bnn_fail(code, msg); }
#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) return fail(BNN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct bnn_plan {   // (the C ABI's opaque type: global namespace)
    using Tables = bnn::Tables;
    using GenArch = bnn::GenArch;
    bnn_arch arch;
    bool megno = false;  // arch.fix_megno
    bool v50net = false; // the pretrained ensemble's network: 41 -> 40 -> 40 -> 20 / 40 (42) -> 40 -> 40 -> 2 (register-resident kernels)
    int d = bnn::D;      // length of the flat parameter vector
    Tables tab[2];  // [0] = arch mask, [1] = all 41 columns (noisy forward)      (v50net only)
    int16_t* d_f2 = nullptr;   // regress_nn fragment gather table
    int16_t* d_f4 = nullptr;   // feature_nn weight-register table (4x4x1 path) for the plan's mask
    int16_t* d_f4n = nullptr;  // ... with every column live (noisy forward)
    float* d_rcp = nullptr;    // [RCP_N] 1/(i+1)
    GenArch gen;               // generic engine: every plan has one (the v50 network falls back to it for T % 4 != 0 or T < 8)
    GenArch* d_gen = nullptr;
    // specialised forms of the generic engine, compiled at run time for this network (bnn_spec_source / bnn_plan_attach_spec): [noisy]
    GenArch spec_gen[2];
    hipModule_t spec_mod[2] = {nullptr, nullptr};
    hipFunction_t spec_fn[2] = {nullptr, nullptr};
    bool emb[2] = {false, false};   // the pretrained network's specialised forms compiled into the library (bnn_fwd_v50spec.hip) apply: [noisy]
    GenArch emb_gen[2];
    int device = 0;
};

namespace bnn {
// bnn_stats (the C struct) -> the kernels' parameter block, validated (bnn_ops_stats.hip)
int stats_params(const bnn_stats* st, StatsParams* sp);
// the two float64 constants of the draw rounded as the reference's tensor ops round them (:834-835); K is checked against kmax: MAXK_DRAW
// for the draw kernel, MAXK for the forward kernels' in-prologue draw (bnn_ops_draw.hip)
int draw_consts(int K, float scale, float* c1, float* c2, int kmax);
}  // namespace bnn
