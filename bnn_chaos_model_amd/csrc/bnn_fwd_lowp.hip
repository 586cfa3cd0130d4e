// bnn_fwd_lowp.hip -- the opt-in reduced-precision forward kernels (bf16 matrix pipe; BASELINE.json configs[4] sweep).
#include "bnn_lowp.hip.h"

namespace bnn {
hipError_t launch_fwd_lowp(int nsplit, unsigned nblk, hipStream_t st, const FwdParams& p) {
    switch (nsplit) {
        case 1: return launch_lowp_form<1>(nblk, st, p);
        case 2: return launch_lowp_form<2>(nblk, st, p);
        case 3: return launch_lowp_form<3>(nblk, st, p);
        default: return hipErrorInvalidValue;
    }
}
}  // namespace bnn
