#!/bin/bash
# Prints per-kernel register/LDS/spill numbers for bnn_kernels.hip (compile-only, device side).
cd "$(dirname "$0")"
mkdir -p /tmp/bk
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -x hip bnn_kernels.hip -c \
  --cuda-device-only -save-temps=obj -o /tmp/bk/k.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  grep -E "Function Name|VGPRs:|VGPRs Spill|SGPRs:|ScratchSize|Occupancy" |
  sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | awk '/Function Name/{printf "\n%s ", $3} !/Function Name/{printf "| %s ", $0}'
echo
