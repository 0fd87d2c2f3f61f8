#!/bin/bash
# usage: build_ref.sh <git-ref> <name>  -> gpurun_ab/libcalico_hip_<name>.so built from that ref's sources
set -e
REF=$1; NAME=$2
TMP=/tmp/calico_ref_$NAME; rm -rf $TMP; mkdir -p $TMP
cd /root/repo
git archive $REF calico_amd/csrc include | tar -x -C $TMP
cd $TMP/calico_amd/csrc
for f in calico_hip.cpp eval_kernels.hip solve_kernels.hip bcr_kernels.hip fit_kernels.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c $f -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/gpurun_ab/libcalico_hip_$NAME.so *.o -ldl
echo built $NAME from $REF
