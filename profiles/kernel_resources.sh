#!/bin/bash
# Register / scratch / LDS use of every kernel of a source file, from the compiler's own remarks:
#   bash profiles/kernel_resources.sh calico_amd/csrc/bcr_kernels.hip [filter]
# The flags are the BUILD's: the common ones and the per-file ones of __graft_entry__.py (eval_kernels.hip is compiled with
# -mllvm -amdgpu-mfma-vgpr-form=1; without it the remarks describe a kernel the build never produces).
F=$1; PAT=${2:-.}
REPO=$(cd "$(dirname "$0")/.." && pwd)
FLAGS=$(cd "$REPO" && python3 -c "import os, sys; import __graft_entry__ as g; f = [x for x in g.HIP_FLAGS if x != '-fPIC']; print(' '.join(f + g.HIP_FILE_FLAGS.get(os.path.basename(sys.argv[1]), [])))" "$F")
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -c "$F" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       /VGPRs:/ && !/AGPRs|Spill/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ .*/,"",v)}
       /AGPRs:/ {a=$0; sub(/.*AGPRs: /,"",a); sub(/ .*/,"",a)}
       /ScratchSize/ {s=$0; sub(/.*: /,"",s); sub(/ .*/,"",s)}
       /SGPRs Spill:/ {ss=$0; sub(/.*: /,"",ss); sub(/ .*/,"",ss)}
       /VGPRs Spill:/ {vs=$0; sub(/.*: /,"",vs); sub(/ .*/,"",vs)}
       /Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o)}
       /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); printf "%-60.60s vgpr %3s agpr %3s scratch %4s sgpr_spill %3s vgpr_spill %3s occ %s lds %s\n", name, v, a, s, ss, vs, o, l}' | grep -E "$PAT"
