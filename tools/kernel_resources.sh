#!/bin/bash
# Register / spill / LDS use of every kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh k_backward.hip
cd "$(dirname "$0")/../dags_vae_search_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kr_$$.o 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size" |
  sed -e 's/.*remark: [^ ]* //' | paste - - - - - - - - 2>/dev/null | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | tr -s ' \t' ' '
rm -f /tmp/kr_$$.o
