#!/bin/bash
# usage: trace_ab.sh <variant.so> [rounds]  -- rocprofv3 kernel trace of the in-tree library and of gpurun_ab/<variant.so> on the same box,
# alternating; prints the average duration of the WORKING launches of the iteration's kernels (us) for each run.
# (An A/B of iterations/s resolves ~0.3 %; a kernel's head is worth less than that -- the trace resolves 0.05 us per kernel.)
V=$1; R=${2:-2}
REPO=/root/repo
cd /tmp && export TMPDIR=/tmp CALICO_DEV=1
for i in $(seq $R); do for lib in base $V; do
  if [ "$lib" = base ]; then unset CALICO_HIP_LIB; else export CALICO_HIP_LIB=$REPO/gpurun_ab/$lib; fi
  rm -rf /tmp/prof_ab; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_ab -o kt -- python $REPO/bench.py --no-cpu-baseline --repeats 8 > /dev/null 2>&1
  python $REPO/profiles/summarize_rocpd.py /tmp/prof_ab/*.db | python -c "
import sys, csv
rows = list(csv.DictReader(sys.stdin))
want = ['bcr_level_kernelILb1', 'bcr_level_kernelILb0', 'dense_back', 'eval_cells', 'eval_jacobian', 'expand_cells', 'gather_kernel']
out, tot = [], 0.0
for w in want:
    r = [x for x in rows if w in x['Name']]
    v = float(r[0]['AverageWorkingNs']) / 1e3 if r else 0.0
    tot += v; out.append('%s %.2f' % (w.replace('bcr_level_kernelILb1', 'level0').replace('bcr_level_kernelILb0', 'level1'), v))
print('%-28s' % '$lib', ' | '.join(out), '| sum %.2f' % tot)"
done; done
