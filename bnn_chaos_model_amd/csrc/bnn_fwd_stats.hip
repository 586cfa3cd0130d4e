// bnn_fwd_stats.hip -- quiet forward whose tail runs the scripts' post-sampling statistics per evaluation (truncated-normal
// draw, prior resampling: bnn_stats.hip.h) and stores one log10 instability time per evaluation instead of (mu, std).
#include "bnn_forward.hip.h"

namespace bnn {
hipError_t launch_fwd_stats(bool k31, unsigned nblk, hipStream_t st, const FwdParams& p) {
    return k31 ? launch_forward_form<31, false, false, true>(nblk, st, p) : launch_forward_form<F, false, false, true>(nblk, st, p);
}
}  // namespace bnn
