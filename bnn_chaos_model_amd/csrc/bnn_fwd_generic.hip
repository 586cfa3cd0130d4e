// bnn_fwd_generic.hip -- instantiations of the generic forward engine (bnn_generic.hip.h): input quads 11 | 21 (41 | 82 features)
// x activation buckets of 12 / 16 / 24 / 32 quads (layer widths up to 48 / 64 / 96 / 128).
#include "bnn_generic.hip.h"

namespace bnn {
hipError_t launch_fwd_generic(const GenArch& g, unsigned nblk, hipStream_t st, const GenParams& P) {
    const size_t lds = (size_t)g.lds_bytes;
    if (g.fq == 11) {
        switch (g.hq) {
            case 12: return launch_generic_form<11, 12>(nblk, st, P, g.nwaves, lds);
            case 16: return launch_generic_form<11, 16>(nblk, st, P, g.nwaves, lds);
            case 24: return launch_generic_form<11, 24>(nblk, st, P, g.nwaves, lds);
            case 32: return launch_generic_form<11, 32>(nblk, st, P, g.nwaves, lds);
        }
    } else if (g.fq == 21) {
        switch (g.hq) {
            case 12: return launch_generic_form<21, 12>(nblk, st, P, g.nwaves, lds);
            case 16: return launch_generic_form<21, 16>(nblk, st, P, g.nwaves, lds);
            case 24: return launch_generic_form<21, 24>(nblk, st, P, g.nwaves, lds);
            case 32: return launch_generic_form<21, 32>(nblk, st, P, g.nwaves, lds);
        }
    }
    return hipErrorInvalidValue;
}
}  // namespace bnn
