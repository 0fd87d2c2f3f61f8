#!/bin/bash
# builds the in-tree library (product flags) and a CALICO_DEV_TIMING=1 copy under gpurun_ab/ (here, no GPU needed)
set -e
cd /root/repo
python -c "import __graft_entry__ as g; g.build_hip()"
cp calico_amd/libcalico_hip.so gpurun_ab/libcalico_hip_keep.so
CALICO_DEV_TIMING=1 python -c "import __graft_entry__ as g; g.build_hip()"
cp calico_amd/libcalico_hip.so gpurun_ab/libcalico_hip_devtiming.so
cp gpurun_ab/libcalico_hip_keep.so calico_amd/libcalico_hip.so
python -c "import __graft_entry__ as g; g.build_hip()"
