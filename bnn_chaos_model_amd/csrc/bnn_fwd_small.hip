// bnn_fwd_small.hip -- the tile-split launch form of the forward kernel (TSPLIT, bnn_forward.hip.h) for the v50 column mask: what the
// evaluation scripts' per-chunk calls run on (15 .. 3 000 rows under one draw: a grid of a few hundred wave-batches at most).
#include "bnn_forward.hip.h"

namespace bnn {
hipError_t launch_fwd_small(bool fused, unsigned nblk, hipStream_t st, const FwdParams& p) {
    return fused ? launch_forward_form<31, true, false, false, false, false, true>(nblk, st, p)
                 : launch_forward_form<31, false, false, false, false, false, true>(nblk, st, p);
}
}  // namespace bnn
