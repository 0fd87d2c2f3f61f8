#!/bin/bash
# builds profiles/microbench/block_factor.hip here (cross-compile) and runs it on a GPU box
set -e
cd /root/repo/profiles/microbench
mkdir -p bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -I /root/repo/calico_amd/csrc block_factor.hip -o bin/block_factor -save-temps=obj 2>&1 | grep -E "error" -A5 || true
grep -n "ScratchSize\|; NumVgprs:" bin/block_factor-hip-amdgcn-amd-amdhsa-gfx950.s | head
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 300 -- 'timeout 30 profiles/microbench/bin/block_factor' 2>&1 | tail -8
