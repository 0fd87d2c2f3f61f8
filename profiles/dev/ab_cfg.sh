#!/bin/bash
# usage: ab_cfg.sh CONFIG VAR [rounds]
C=$1; V=$2; R=${3:-2}
cd /root/repo
for i in $(seq $R); do for f in 1 0; do
  env $V=$f timeout 120 python bench.py --config $C --steps 100 --warmup 20 --no-cpu-baseline --repeats 9 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg $C $V=$f', round(d['value'],1), d['ms_per_step'])"
done; done
