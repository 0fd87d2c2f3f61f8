#!/bin/bash
# on the GPU box, from the repo root: WRITE_SIZE / FETCH_SIZE per kernel of profiles/microbench/bin/write_size (one PMC pass per counter)
OUT=${1:-gpurun_out/r05}
mkdir -p $OUT; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/ws_$c; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/ws_$c -o ws -- $REPO/profiles/microbench/bin/write_size > /tmp/ws_$c.log 2>&1
done
python3 - <<PY > $REPO/$OUT/r05_write_size_calibration.txt
import csv, collections, glob
print(open('/tmp/ws_WRITE_SIZE.log').read().strip().splitlines()[-1])
for c in ('WRITE_SIZE', 'FETCH_SIZE'):
    f = glob.glob('/tmp/ws_%s/*counter_collection.csv' % c)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == c: agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print('%-11s %-28s per launch (KiB as reported): %s  -> %.1f MB' % (c, k, ' '.join('%.0f' % x for x in v), sum(v) / len(v) * 1024 / 1e6))
PY
cat $REPO/$OUT/r05_write_size_calibration.txt
