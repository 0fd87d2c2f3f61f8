#!/bin/bash
# usage: build_variant.sh <name> <source file of calico_amd/csrc> "<extra flags>"  -> gpurun_ab/libcalico_hip_<name>.so: the working
# tree's library with ONE source recompiled with extra flags (the other objects are the in-tree build's: run build() first)
set -e
NAME=$1; SRC=$2; EXTRA=$3
cd /root/repo
FLAGS=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIP_FLAGS + g.HIP_FILE_FLAGS.get('$SRC', [])))")
B=calico_amd/csrc/build
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c calico_amd/csrc/$SRC -o /tmp/variant_$NAME.o
OBJS=""
for f in $(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIP_SOURCES))"); do
  if [ "$f" = "$SRC" ]; then OBJS="$OBJS /tmp/variant_$NAME.o"; else OBJS="$OBJS $B/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_ab/libcalico_hip_$NAME.so $OBJS -ldl
echo built gpurun_ab/libcalico_hip_$NAME.so
