// bnn_fwd_noisy.hip -- VarModel.forward(noisy_val=True) (spock_reg_model.py:444-450, 486-528): input noise on all 41 columns
// after masking, summary noise before regress_nn; explicit noise tensors or in-kernel Philox.
#include "bnn_forward.hip.h"

namespace bnn {
hipError_t launch_fwd_noisy(unsigned nblk, hipStream_t st, const FwdParams& p) {
    return p.eps_in ? launch_forward_form<F, false, true, false, false, true>(nblk, st, p)
                    : launch_forward_form<F, false, true, false, false, false>(nblk, st, p);
}
}  // namespace bnn
