// bnn_fwd_generic82.hip -- the generic forward engine for 82 features (hparams['include_derivatives'], spock_reg_model.py:346, 358):
// 21 input quads; one wave per SIMD in every bucket (the x rows alone are 84 registers).
#include "bnn_generic.hip.h"

namespace bnn {
hipError_t launch_fwd_generic82(const GenArch& g, unsigned nblk, hipStream_t st, const GenParams& P) {
    const size_t lds = (size_t)g.lds_bytes;
    switch (g.hq) {
        case 12: return launch_generic_form<21, 12, false>(nblk, st, P, g.nwaves, lds);
        case 16: return launch_generic_form<21, 16, false>(nblk, st, P, g.nwaves, lds);
        case 24: return launch_generic_form<21, 24, false>(nblk, st, P, g.nwaves, lds);
        case 32: return launch_generic_form<21, 32, false>(nblk, st, P, g.nwaves, lds);
    }
    return hipErrorInvalidValue;
}
}  // namespace bnn
