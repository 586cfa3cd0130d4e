// bnn_fwd_generic.hip -- the generic forward engine (bnn_generic.hip.h) for 41 features: activation buckets of 12 / 16 / 24 / 32 quads
// (layer widths up to 48 / 64 / 96 / 128); the two narrow buckets also in their eight-wave (256-register) form.
#include "bnn_generic.hip.h"

namespace bnn {
hipError_t launch_fwd_generic82(const GenArch& g, unsigned nblk, hipStream_t st, const GenParams& P);   // bnn_fwd_generic82.hip

hipError_t launch_fwd_generic(const GenArch& g, unsigned nblk, hipStream_t st, const GenParams& P) {
    const size_t lds = (size_t)g.lds_bytes;
    if (g.fq == 21) return launch_fwd_generic82(g, nblk, st, P);
    if (g.fq != 11) return hipErrorInvalidValue;
    const bool w8 = g.nwaves == 8;
    switch (g.hq) {
        case 12: return w8 ? launch_generic_form<11, 12, true>(nblk, st, P, g.nwaves, lds) : launch_generic_form<11, 12, false>(nblk, st, P, g.nwaves, lds);
        case 16: return w8 ? launch_generic_form<11, 16, true>(nblk, st, P, g.nwaves, lds) : launch_generic_form<11, 16, false>(nblk, st, P, g.nwaves, lds);
        case 24: return launch_generic_form<11, 24, false>(nblk, st, P, g.nwaves, lds);
        case 32: return launch_generic_form<11, 32, false>(nblk, st, P, g.nwaves, lds);
    }
    return hipErrorInvalidValue;
}
}  // namespace bnn
