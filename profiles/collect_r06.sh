#!/bin/bash
# Round-6 measurement set, run on the GPU box from the repo root:  bash profiles/collect_r06.sh [quick]
# Writes under gpurun_out/r06/ (summaries) and gpurun_out/r06/raw/ (the rocprofv3 databases / counter CSVs they come
# from, kept so that every figure in profiles/r06_* can be traced); the summaries are then copied into profiles/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06
RAW=$OUT/raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
# 1. the bench line as the driver runs it (with the CPU baseline)
timeout 900 python $REPO/bench.py 2>$OUT/bench_stderr.log | tail -1 > $OUT/r06_bench.json
# 2. kernel trace of the same command (without the CPU baseline leg)
rm -rf /tmp/prof_kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $REPO/bench.py --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 > $OUT/r06_bench_under_rocprof.json
python $REPO/profiles/summarize_rocpd.py /tmp/prof_kt/*.db $OUT/r06_kernel_stats.csv
python $REPO/profiles/timeline_gaps.py /tmp/prof_kt/*.db > $OUT/r06_timeline_gaps.txt
python $REPO/profiles/iteration_trace.py /tmp/prof_kt/*.db bcr_level_kernelILb1 > $OUT/r06_iteration_trace.txt
python $REPO/profiles/solve_boundary_gaps.py /tmp/prof_kt/*.db > $OUT/r06_solve_boundary.txt
cp /tmp/prof_kt/*.db $RAW/kernel_trace.db 2>/dev/null; cp /tmp/prof_kt/*stats*.csv $RAW/ 2>/dev/null
# 3. HBM traffic: one PMC pass per counter
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 2 --repeats 1 > /dev/null 2>&1
  cp $(ls /tmp/pmc_$c/*counter_collection.csv | head -1) $RAW/pmc_${c}_counter_collection.csv
done
python $REPO/profiles/hbm_traffic_from_pmc.py $RAW/pmc_FETCH_SIZE_counter_collection.csv $RAW/pmc_WRITE_SIZE_counter_collection.csv $OUT/r06_pmc_hbm_by_kernel.csv $OUT/hbm_traffic.json
# 4. FP64 / matrix-core utilisation: SQ counters in a pass of their own, the chip's active cycles in another
rm -rf /tmp/pmc_fp64; timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d /tmp/pmc_fp64 -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 2 --repeats 1 > /dev/null 2>&1
cp $(ls /tmp/pmc_fp64/*counter_collection.csv | head -1) $RAW/pmc_fp64_counter_collection.csv
rm -rf /tmp/pmc_gui; timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_gui -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 2 --repeats 1 > /dev/null 2>&1
GUI=$(ls /tmp/pmc_gui/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$GUI" ]; then cp $GUI $RAW/pmc_gui_counter_collection.csv; fi
python $REPO/profiles/fp64_from_pmc.py $RAW/pmc_fp64_counter_collection.csv $OUT/r06_kernel_stats.csv $OUT/r06_pmc_fp64_by_kernel.csv $OUT/fp64_utilisation.json $( [ -n "$GUI" ] && echo $RAW/pmc_gui_counter_collection.csv )
if [ "${1:-}" = "quick" ]; then ls -la $OUT; exit 0; fi
# 5. the other configurations (parity-test cases and context shapes, not bench lines)
for c in 1 2; do timeout 600 python $REPO/bench.py --config $c --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r06_config${c}.json; done
timeout 900 python $REPO/bench.py --config 4 --steps 100 --warmup 20 --repeats 3 --tagging-passes 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r06_config4.json
timeout 600 python $REPO/bench.py --config 5 --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r06_shape_config3_50hz_knots.json
timeout 900 python $REPO/bench.py --config 6 --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r06_shape_notebook_run.json
timeout 300 python $REPO/bench.py --force-collective --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 > $OUT/r06_collective_1gpu.json
# 6. kernel-level evidence for configs[4] (trace, HBM and FP64 counters in passes of their own)
cd $REPO && bash profiles/collect_config.sh r06 4 > $OUT/collect_config4.log 2>&1
# 7. WRITE_SIZE / FETCH_SIZE against known byte counts in the evaluation chain's access patterns
cd $REPO && bash profiles/microbench/run_write_size.sh gpurun_out/r06 > /dev/null 2>&1
# 8. what one rank of an N-GPU run spends per iteration (shards of 1/2/4/8 measured on this one GPU)
cd $REPO && timeout 900 python profiles/shard_scaling_model.py 3 4 > $OUT/r06_shard_scaling_model.json 2>/dev/null
# 9. same-box A/B of this round's changes against the round's first commit (gpurun_ab/libcalico_hip_base.so, built by profiles/dev/build_ref.sh)
if [ -f $REPO/gpurun_ab/libcalico_hip_base.so ]; then cd $REPO && bash profiles/dev/ab.sh libcalico_hip_base.so 3 > $OUT/r06_round_ab.txt 2>&1; fi
if [ -x $REPO/profiles/microbench/bin/block_factor ]; then timeout 60 $REPO/profiles/microbench/bin/block_factor > $OUT/r06_block_elimination_microbench.txt 2>&1; fi
ls -la $OUT
