#!/bin/bash
# Round-1 measurement set, run on the GPU box from the repo root:  bash profiles/collect_r01.sh
# Writes under gpurun_out/final/; the summaries are then copied into profiles/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line as the driver runs it (with the CPU baseline)
timeout 900 python $REPO/bench.py 2>$OUT/bench_stderr.log | tail -1 > $OUT/r01_bench.json
# 2. kernel trace of the same command (without the CPU baseline leg)
rm -rf /tmp/prof_kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $REPO/bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r01_bench_under_rocprof.json
python $REPO/profiles/summarize_rocpd.py /tmp/prof_kt/*.db $OUT/r01_kernel_stats.csv
python $REPO/profiles/timeline_gaps.py /tmp/prof_kt/*.db > $OUT/r01_timeline_gaps.txt
# 3. HBM traffic: one PMC pass per counter
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 2 > /dev/null 2>&1
done
python $REPO/profiles/hbm_traffic_from_pmc.py $(ls /tmp/pmc_FETCH_SIZE/*counter_collection.csv | head -1) $(ls /tmp/pmc_WRITE_SIZE/*counter_collection.csv | head -1) $OUT/r01_pmc_hbm_by_kernel.csv $OUT/hbm_traffic.json
# 4. the other configurations (parity-test cases, not bench lines)
for c in 1 2 4; do timeout 600 python $REPO/bench.py --config $c --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r01_config${c}.json; done
timeout 300 python $REPO/bench.py --force-collective --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r01_collective_1gpu.json
ls -la $OUT
