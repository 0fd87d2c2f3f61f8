#!/bin/bash
# usage: ab_multi.sh ROUNDS variant1.so variant2.so ...  -- alternates the in-tree library and every variant of gpurun_ab/ on one box
R=$1; shift
cd /root/repo
export CALICO_DEV=1
for i in $(seq $R); do
  for lib in base "$@"; do
    if [ "$lib" = base ]; then unset CALICO_HIP_LIB; else export CALICO_HIP_LIB=/root/repo/gpurun_ab/$lib; fi
    timeout 120 python bench.py --no-cpu-baseline --repeats 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), d['ms_per_step'])"
  done
done
