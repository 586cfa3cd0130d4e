// bnn_fwd_k31.hip -- forward kernel for the v50 column mask (31 live input columns): the headline configuration.
#include "bnn_forward.hip.h"

namespace bnn {
hipError_t launch_fwd_k31(bool fused, unsigned nblk, hipStream_t st, const FwdParams& p) {
    return fused ? launch_forward_form<31, true, false, false>(nblk, st, p) : launch_forward_form<31, false, false, false>(nblk, st, p);
}
}  // namespace bnn
