#!/bin/bash
# kernel resource usage of one .hip file: VGPRs / scratch / LDS / occupancy
cd "$(dirname "$0")/../cs231-capsule-yolo-traffic-sign-detection_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -munsafe-fp-atomics -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|  VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - - \
 | sed -e 's/Function Name: _ZN12_GLOBAL__N_1//' -e 's/ScratchSize \[bytes\/lane\]/Scratch/' -e 's/Occupancy \[waves\/SIMD\]/Occ/' -e 's/LDS Size \[bytes\/block\]/LDS/'
