#!/bin/bash
# Round-4 measurement set, run on the GPU box from the repo root:  bash profiles/collect_r04.sh [quick]
# Writes under gpurun_out/r04/ (summaries) and gpurun_out/r04/raw/ (the rocprofv3 databases / counter CSVs they come
# from, kept so that every figure in profiles/r04_* can be traced); the summaries are then copied into profiles/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04
RAW=$OUT/raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
# 1. the bench line as the driver runs it (with the CPU baseline)
timeout 900 python $REPO/bench.py 2>$OUT/bench_stderr.log | tail -1 > $OUT/r04_bench.json
# 2. kernel trace of the same command (without the CPU baseline leg)
rm -rf /tmp/prof_kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $REPO/bench.py --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 > $OUT/r04_bench_under_rocprof.json
python $REPO/profiles/summarize_rocpd.py /tmp/prof_kt/*.db $OUT/r04_kernel_stats.csv
python $REPO/profiles/timeline_gaps.py /tmp/prof_kt/*.db > $OUT/r04_timeline_gaps.txt
python $REPO/profiles/iteration_trace.py /tmp/prof_kt/*.db bcr_level_kernelILb1 > $OUT/r04_iteration_trace.txt
python $REPO/profiles/solve_boundary_gaps.py /tmp/prof_kt/*.db > $OUT/r04_solve_boundary.txt
cp /tmp/prof_kt/*.db $RAW/kernel_trace.db 2>/dev/null; cp /tmp/prof_kt/*stats*.csv $RAW/ 2>/dev/null
# 3. HBM traffic: one PMC pass per counter
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 2 --repeats 1 > /dev/null 2>&1
  cp $(ls /tmp/pmc_$c/*counter_collection.csv | head -1) $RAW/pmc_${c}_counter_collection.csv
done
python $REPO/profiles/hbm_traffic_from_pmc.py $RAW/pmc_FETCH_SIZE_counter_collection.csv $RAW/pmc_WRITE_SIZE_counter_collection.csv $OUT/r04_pmc_hbm_by_kernel.csv $OUT/hbm_traffic.json
# 4. FP64 / matrix-core utilisation: SQ counters in a pass of their own, the chip's active cycles in another
rm -rf /tmp/pmc_fp64; timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d /tmp/pmc_fp64 -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 2 --repeats 1 > /dev/null 2>&1
cp $(ls /tmp/pmc_fp64/*counter_collection.csv | head -1) $RAW/pmc_fp64_counter_collection.csv
rm -rf /tmp/pmc_gui; timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_gui -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 2 --repeats 1 > /dev/null 2>&1
GUI=$(ls /tmp/pmc_gui/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$GUI" ]; then cp $GUI $RAW/pmc_gui_counter_collection.csv; fi
python $REPO/profiles/fp64_from_pmc.py $RAW/pmc_fp64_counter_collection.csv $OUT/r04_kernel_stats.csv $OUT/r04_pmc_fp64_by_kernel.csv $OUT/fp64_utilisation.json $( [ -n "$GUI" ] && echo $RAW/pmc_gui_counter_collection.csv )
if [ "${1:-}" = "quick" ]; then ls -la $OUT; exit 0; fi
# 5. the other configurations (parity-test cases and context shapes, not bench lines)
for c in 1 2; do timeout 600 python $REPO/bench.py --config $c --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r04_config${c}.json; done
timeout 900 python $REPO/bench.py --config 4 --steps 100 --warmup 20 --repeats 3 --tagging-passes 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r04_config4.json
timeout 600 python $REPO/bench.py --config 5 --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r04_shape_config3_50hz_knots.json
timeout 900 python $REPO/bench.py --config 6 --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r04_shape_notebook_run.json
timeout 300 python $REPO/bench.py --force-collective --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 > $OUT/r04_collective_1gpu.json
# 6. A/B of this round's block elimination (same box): CALICO_ELIM=panel is the block factorisation of rounds 1-3
for v in panel mfma panel mfma panel mfma; do CALICO_ELIM=$v timeout 300 python $REPO/bench.py --no-cpu-baseline --repeats 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('CALICO_ELIM=$v', round(d['value'],1), 'it/s', d['ms_per_step'], 'ms/iteration')"; done > $OUT/r04_elim_ab.txt
for c in 1 2 4; do for v in panel mfma; do CALICO_ELIM=$v timeout 300 python $REPO/bench.py --config $c --no-cpu-baseline --repeats 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[$c] CALICO_ELIM=$v', round(d['value'],1), 'it/s')"; done; done >> $OUT/r04_elim_ab.txt
# 6b. A/B of the round's second half (same box): cell workgroups (CALICO_FUSE_EXPAND=0: records + expansion launch + row cells),
#     the end-of-solve hint (CALICO_PREDICT_END=0), the device arena (CALICO_ARENA=0), the XCD-aware gather (CALICO_GATHER_XCD=0)
for sw in CALICO_FUSE_EXPAND CALICO_PREDICT_END CALICO_ARENA CALICO_GATHER_XCD; do for r in 1 2 3; do for v in 0 1; do env $sw=$v timeout 300 python $REPO/bench.py --no-cpu-baseline --repeats 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$sw=$v', round(d['value'],1), 'it/s', d['ms_per_step'], 'ms/iteration')"; done; done; done > $OUT/r04_second_half_ab.txt
for c in 1 2 4; do for v in 0 1; do CALICO_FUSE_EXPAND=$v timeout 300 python $REPO/bench.py --config $c --no-cpu-baseline --repeats 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[$c] CALICO_FUSE_EXPAND=$v', round(d['value'],1), 'it/s')"; done; done >> $OUT/r04_second_half_ab.txt
if [ -x $REPO/profiles/microbench/bin/any_order ]; then timeout 90 $REPO/profiles/microbench/bin/any_order > $OUT/r04_any_order_launch_microbench.txt 2>&1; fi
if [ -x $REPO/profiles/microbench/bin/first_touch ]; then timeout 90 $REPO/profiles/microbench/bin/first_touch > $OUT/r04_first_touch_microbench.txt 2>&1; fi
if [ -x $REPO/profiles/microbench/bin/icache_cold ]; then timeout 90 $REPO/profiles/microbench/bin/icache_cold > $OUT/r04_instruction_fetch_microbench.txt 2>&1; fi
# 6c. the heads of the tree levels / the reduced solve (third part of the round): the in-tree build against HEAD~n is not
#     kept as a switch (the changes are structural); profiles/r04_heads_ab.txt holds the same-box runs of the variants
# 7. the 32x32 block elimination alone on a CU (microbenchmark; built by profiles/microbench/run_block_factor.sh)
if [ -x $REPO/profiles/microbench/bin/block_factor ]; then timeout 60 $REPO/profiles/microbench/bin/block_factor > $OUT/r04_block_elimination_microbench.txt 2>&1; fi
ls -la $OUT
