#!/bin/bash
# per-kernel AVERAGE durations (kernel trace over a short bench run) for each value of an env switch, alternating on one box
# usage: trace_avg.sh VAR [rounds] [extra bench args]
V=$1; R=${2:-2}; shift; shift
cd /tmp && export TMPDIR=/tmp
for i in $(seq $R); do for f in 1 0; do
  rm -rf /tmp/prof_ta; env $V=$f timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ta -o kt -- python /root/repo/bench.py --no-cpu-baseline --repeats 5 "$@" > /dev/null 2>&1
  python /root/repo/profiles/summarize_rocpd.py /tmp/prof_ta/*.db /tmp/ta.csv > /dev/null 2>&1
  echo "$V=$f"; python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/ta.csv')):
    n=r['Name']
    if any(k in n for k in ('bcr_level','dense_back','eval_cells','gather_kernelI')): print('   %-46s %5s %9.1f' % (n[:46], r['Calls'], float(r['AverageWorkingNs'])))
PY
done; done
