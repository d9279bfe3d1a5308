#!/bin/bash
# Register / scratch / LDS use of the kernels of one source file (device-side compile only): tools/kernel_resources.sh frame_head_lp.hip [name filter]
cd "$(dirname "$0")/../genefaceplusplus_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | grep -E "Function Name|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size" |
    sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - - - | grep -E "${2:-.}"
