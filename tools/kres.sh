#!/bin/bash
# compact per-kernel resource usage of one translation unit: tools/kres.sh k_fused.hip -DODINN_LM=0 [grep-pattern]
cd "$(dirname "$0")/../odinn.jl_amd/csrc"
src=$1; def=$2; pat=${3:-.}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $def -Rpass-analysis=kernel-resource-usage -c $src -o /tmp/kres.o 2>&1 \
 | grep -E "Function Name|Name:|VGPRs:|VGPRs Spill|ScratchSize|LDS Size|SGPRs:" \
 | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - | c++filt | sed -E 's/\(odinn::Pools.*\)//' | grep -E "$pat"
