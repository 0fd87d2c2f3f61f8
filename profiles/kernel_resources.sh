#!/bin/bash
# Register / scratch / LDS use of every kernel of a source file, from the compiler's own remarks:
#   bash profiles/kernel_resources.sh calico_amd/csrc/bcr_kernels.hip [filter]
F=$1; PAT=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -c "$F" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       /VGPRs:/ && !/AGPRs|Spill/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ .*/,"",v)}
       /AGPRs:/ {a=$0; sub(/.*AGPRs: /,"",a); sub(/ .*/,"",a)}
       /ScratchSize/ {s=$0; sub(/.*: /,"",s); sub(/ .*/,"",s)}
       /SGPRs Spill:/ {ss=$0; sub(/.*: /,"",ss); sub(/ .*/,"",ss)}
       /VGPRs Spill:/ {vs=$0; sub(/.*: /,"",vs); sub(/ .*/,"",vs)}
       /Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o)}
       /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); printf "%-60.60s vgpr %3s agpr %3s scratch %4s sgpr_spill %3s vgpr_spill %3s occ %s lds %s\n", name, v, a, s, ss, vs, o, l}' | grep -E "$PAT"
