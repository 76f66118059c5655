#!/bin/bash
# rebuild libslow5gpu.so from the repo root; prints the register usage of the fused kernel
cd "$(dirname "$0")/.." || exit 1
python -c "from slow5tools_amd import build; build.build()" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c slow5tools_amd/csrc/kernels.hip -o /tmp/k_probe.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "Function Name: _Z14k_encode_fused" | grep -E "VGPRs:|Occupancy|ScratchSize" | sed 's/.*remark: *//'
ls -la slow5tools_amd/libslow5gpu.so
