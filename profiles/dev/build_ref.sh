#!/bin/bash
# usage: build_ref.sh <git-ref> <name>  -> gpurun_ab/libcalico_hip_<name>.so built from that ref's sources with that ref's
# build flags (common + per-file: __graft_entry__.HIP_FLAGS / HIP_FILE_FLAGS of the ref)
set -e
REF=$1; NAME=$2
TMP=/tmp/calico_ref_$NAME; rm -rf $TMP; mkdir -p $TMP
cd /root/repo
git archive $REF calico_amd/csrc include __graft_entry__.py | tar -x -C $TMP
cd $TMP
python3 - <<PY
import subprocess, os, sys
sys.path.insert(0, "$TMP")
import __graft_entry__ as g
from concurrent.futures import ThreadPoolExecutor
csrc = os.path.join("$TMP", "calico_amd", "csrc")
def cc(f):
    subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + getattr(g, "HIP_FILE_FLAGS", {}).get(f, []) + ["-c", os.path.join(csrc, f), "-o", os.path.join(csrc, f + ".o")])
    return os.path.join(csrc, f + ".o")
with ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(cc, g.HIP_SOURCES))
subprocess.check_call([g.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/root/repo/gpurun_ab/libcalico_hip_$NAME.so"] + objs + ["-ldl"])
PY
echo built $NAME from $REF
