// bnn_fwd_megno.hip -- forward kernel forms for hparams['fix_megno'] = True (spock_reg_model.py:360-362, 480-491, 509-510): 42-wide
// summary, d = 7665.  No pretrained checkpoint uses the flag; the forms exist so that the surface covers the reference's branch.
#include "bnn_forward.hip.h"

namespace bnn {
hipError_t launch_fwd_megno(bool k31, bool fused, bool noisy, unsigned nblk, hipStream_t st, const FwdParams& p) {
    if (noisy) return p.eps_in ? launch_forward_form<F, false, true, false, true, true>(nblk, st, p)
                               : launch_forward_form<F, false, true, false, true, false>(nblk, st, p);
    if (k31) return fused ? launch_forward_form<31, true, false, false, true>(nblk, st, p) : launch_forward_form<31, false, false, false, true>(nblk, st, p);
    return fused ? launch_forward_form<F, true, false, false, true>(nblk, st, p) : launch_forward_form<F, false, false, false, true>(nblk, st, p);
}
}  // namespace bnn
