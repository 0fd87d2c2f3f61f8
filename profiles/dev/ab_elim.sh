#!/bin/bash
# usage: ab_elim.sh [rounds] [extra bench args]  -- alternates CALICO_ELIM=panel (rounds 1-3 block factorisation) and the default on one box
R=${1:-3}; shift
cd /root/repo
for i in $(seq $R); do
  for v in panel mfma; do
    CALICO_ELIM=$v timeout 120 python bench.py --no-cpu-baseline --repeats 60 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), d['ms_per_step'])"
  done
done
