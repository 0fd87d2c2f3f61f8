#!/bin/bash
# kernel trace of a short bench run on the GPU box: per-kernel averages + one iteration's trace -> gpurun_out/dev_trace/
# usage: trace.sh [extra bench args]   (env, e.g. CALICO_ELIM=panel, is passed through)
REPO=/root/repo
OUT=$REPO/gpurun_out/dev_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dev; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_dev -o kt -- python $REPO/bench.py --no-cpu-baseline --repeats 5 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('under rocprof:', round(d['value'],1), 'it/s')"
python $REPO/profiles/summarize_rocpd.py /tmp/prof_dev/*.db $OUT/kernel_stats.csv
python $REPO/profiles/iteration_trace.py /tmp/prof_dev/*.db bcr_level_kernelILb1 > $OUT/iteration_trace.txt
cut -d, -f1,2,9 $OUT/kernel_stats.csv | head -12
cat $OUT/iteration_trace.txt
