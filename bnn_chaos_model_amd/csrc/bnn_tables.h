// bnn_tables.h -- host-side construction of the MFMA operand gather tables.
#pragma once
#include <cstdint>
#include <vector>

#include "bnn_layout.h"

namespace bnn {

// Entry [f * 64 + lane] = index into the LDS-resident flat parameter vector (ZERO_IDX for padding)
// that lane `lane` loads into fragment register f.
struct Tables {
    std::vector<int16_t> f2;   // Lay::NF2 * 64: regress_nn fragments + C-init biases
    int kin4;                  // layer-1 inputs of the 4x4x1 kernel: 31 (v50 mask) or 41
    std::vector<int16_t> f4;   // WR<kin4>::NR * 64: feature_nn weight registers of the 4x4x1 kernel (entry [R * 64 + lane])
    std::vector<int32_t> order[6];  // accumulation order per Linear layer (input index or -1 = bias)
};

// zero_mask: columns whose weights are dropped (their x is zeroed by the reference).
// all_columns: keep every column's weight (noisy forward: masked columns carry pure noise).
// megno: hparams['fix_megno'] layout (42-wide summary, d = 7665; bnn_layout.h, Lay<true>).
Tables build_tables(uint64_t zero_mask, bool all_columns, bool megno = false);

}  // namespace bnn
