#!/bin/bash
# usage: ab_env.sh VAR [rounds] -- alternates VAR=1 / VAR=0 on the same box
V=$1; R=${2:-3}
cd /root/repo
for i in $(seq $R); do for f in 1 0; do
  env $V=$f timeout 120 python bench.py --no-cpu-baseline --repeats 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=$f', round(d['value'],1), d['ms_per_step'])"
done; done
