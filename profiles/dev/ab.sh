#!/bin/bash
# usage: ab.sh <variant.so> [rounds]  -- alternates base / variant on the same box
V=$1; R=${2:-3}
cd /root/repo
export CALICO_DEV=1      # (calico_amd/_capi.py honours CALICO_HIP_LIB only then)
for i in $(seq $R); do
  for lib in base $V; do
    if [ "$lib" = base ]; then unset CALICO_HIP_LIB; else export CALICO_HIP_LIB=/root/repo/gpurun_ab/$lib; fi
    timeout 120 python bench.py --no-cpu-baseline --repeats 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), d['ms_per_step'])"
  done
done
