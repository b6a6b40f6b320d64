#!/bin/bash
# Rebuild libbepuhip.so with the kernel resource usage report (developer helper; build.py is the product build).
cd "$(dirname "$0")/../bepuphysics2_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Xarch_device -fno-slp-vectorize -fPIC -shared -Wno-unused-result -Wno-unused-value -Wno-array-bounds \
  -Rpass-analysis=kernel-resource-usage -o libbepuhip.so bepuhip.hip 2>&1 | grep -E "error|Function Name|VGPRs:|Scratch|VGPRs Spill|Occupancy" | sed 's/.*remark: //; s/\[-Rpass.*//' | cut -c1-110
