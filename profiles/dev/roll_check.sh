#!/bin/bash
# same-box check of a level-kernel variant behind an env switch: bit-identity (bitwise.py), A/B of the switch on bench.py, the
# dev-timing anatomy and the kernel trace with it on.   usage: roll_check.sh VAR [tag]
V=${1:-CALICO_ROLL}; T=${2:-roll}
O=gpurun_out/r06a; mkdir -p $O
(echo $V=0; env $V=0 timeout 200 python profiles/dev/bitwise.py; echo $V=1; env $V=1 timeout 200 python profiles/dev/bitwise.py) 2>&1 | grep -v amdgpu.ids > $O/bitwise_$T.txt; cat $O/bitwise_$T.txt
bash profiles/dev/ab_env.sh $V 3 2>&1 | tee $O/ab_$T.txt
(export CALICO_DEV=1 CALICO_HIP_LIB=/root/repo/gpurun_ab/libcalico_hip_devtiming.so; env $V=1 CALICO_KERNEL_TIMING=4 timeout 120 python profiles/dev/devrun.py 3 3 > $O/anat_$T.txt 2>&1)
env $V=1 bash profiles/dev/trace.sh > $O/trace_$T.txt 2>&1; tail -n 7 $O/trace_$T.txt
