// bnn_fwd_lowp.hip -- the opt-in reduced-precision forward kernels (bf16 / half matrix pipe; BASELINE.json configs[4] sweep).
#include "bnn_lowp.hip.h"

namespace bnn {
hipError_t launch_fwd_lowp(int precision, unsigned nblk, hipStream_t st, const FwdParams& p) {
    switch (precision) {  // bnn_precision of include/bnn_chaos_hip.h
        case 1: return launch_lowp_form<1, false>(nblk, st, p);  // bf16
        case 2: return launch_lowp_form<2, false>(nblk, st, p);  // bf16 x3
        case 3: return launch_lowp_form<3, false>(nblk, st, p);  // bf16 x6
        case 4: return launch_lowp_form<1, true>(nblk, st, p);   // f16
        case 5: return launch_lowp_form<2, true>(nblk, st, p);   // f16 x3
        default: return hipErrorInvalidValue;
    }
}
}  // namespace bnn
