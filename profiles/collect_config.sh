#!/bin/bash
# Kernel-level evidence for ONE configuration (bench.py --config C), run on the GPU box from the repo root:
#   bash profiles/collect_config.sh <round tag, e.g. r05> <config index> [bench.py flags...]
# Writes gpurun_out/<tag>/<tag>_config<C>_{kernel_stats.csv,iteration_trace.txt,pmc_hbm_by_kernel.csv,pmc_fp64_by_kernel.csv}
# and the bench line <tag>_config<C>.json; raw rocprofv3 output under gpurun_out/<tag>/raw_config<C>/.
# Counters are collected in passes of their own with --kernel-trace only (never with the hip / hsa trace domains).
set -u
TAG=$1; CFG=$2; shift 2
EXTRA="$*"
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
RAW=$OUT/raw_config$CFG
P=$OUT/${TAG}_config${CFG}
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --config $CFG --no-cpu-baseline $EXTRA"
timeout 900 $BENCH --steps 100 --warmup 20 --repeats 3 2>/dev/null | tail -1 > $P.json
rm -rf /tmp/prof_kt$CFG; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt$CFG -o kt -- $BENCH --steps 40 --warmup 10 --repeats 2 > /dev/null 2>&1
python $REPO/profiles/summarize_rocpd.py /tmp/prof_kt$CFG/*.db ${P}_kernel_stats.csv
python $REPO/profiles/iteration_trace.py /tmp/prof_kt$CFG/*.db bcr_level_kernelILb1 > ${P}_iteration_trace.txt
cp /tmp/prof_kt$CFG/*.db $RAW/kernel_trace.db 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- $BENCH --steps 16 --warmup 2 --repeats 1 > /dev/null 2>&1
  cp $(ls /tmp/pmc_$c/*counter_collection.csv | head -1) $RAW/pmc_${c}_counter_collection.csv
done
python $REPO/profiles/hbm_traffic_from_pmc.py $RAW/pmc_FETCH_SIZE_counter_collection.csv $RAW/pmc_WRITE_SIZE_counter_collection.csv ${P}_pmc_hbm_by_kernel.csv $RAW/hbm_traffic.json
rm -rf /tmp/pmc_fp64; timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d /tmp/pmc_fp64 -o pmc -- $BENCH --steps 16 --warmup 2 --repeats 1 > /dev/null 2>&1
cp $(ls /tmp/pmc_fp64/*counter_collection.csv | head -1) $RAW/pmc_fp64_counter_collection.csv
python $REPO/profiles/fp64_from_pmc.py $RAW/pmc_fp64_counter_collection.csv ${P}_kernel_stats.csv ${P}_pmc_fp64_by_kernel.csv $RAW/fp64_utilisation.json
ls -la $OUT | grep config$CFG
