// bnn_fwd_k41.hip -- forward kernel for any other column mask: whole 41-column rows, zero weights on the masked columns.
#include "bnn_forward.hip.h"

namespace bnn {
hipError_t launch_fwd_k41(bool fused, unsigned nblk, hipStream_t st, const FwdParams& p) {
    return fused ? launch_forward_form<F, true, false, false>(nblk, st, p) : launch_forward_form<F, false, false, false>(nblk, st, p);
}
}  // namespace bnn
