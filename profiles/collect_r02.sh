#!/bin/bash
# Round-2 measurement set, run on the GPU box from the repo root:  bash profiles/collect_r02.sh
# Writes under gpurun_out/r02/; the summaries are then copied into profiles/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line as the driver runs it (with the CPU baseline)
timeout 900 python $REPO/bench.py 2>$OUT/bench_stderr.log | tail -1 > $OUT/r02_bench.json
# 2. kernel trace of the same command (without the CPU baseline leg)
rm -rf /tmp/prof_kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $REPO/bench.py --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 > $OUT/r02_bench_under_rocprof.json
python $REPO/profiles/summarize_rocpd.py /tmp/prof_kt/*.db $OUT/r02_kernel_stats.csv
python $REPO/profiles/timeline_gaps.py /tmp/prof_kt/*.db > $OUT/r02_timeline_gaps.txt
python $REPO/profiles/iteration_trace.py /tmp/prof_kt/*.db bcr_level_kernelILb1 > $OUT/r02_iteration_trace.txt
python $REPO/profiles/solve_boundary_gaps.py /tmp/prof_kt/*.db > $OUT/r02_solve_boundary.txt
# 3. HBM traffic: one PMC pass per counter
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 2 --repeats 1 > /dev/null 2>&1
done
python $REPO/profiles/hbm_traffic_from_pmc.py $(ls /tmp/pmc_FETCH_SIZE/*counter_collection.csv | head -1) $(ls /tmp/pmc_WRITE_SIZE/*counter_collection.csv | head -1) $OUT/r02_pmc_hbm_by_kernel.csv $OUT/hbm_traffic.json
# 4. FP64 / matrix-core utilisation: SQ counters in a pass of their own
rm -rf /tmp/pmc_fp64; timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d /tmp/pmc_fp64 -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 16 --warmup 2 --repeats 1 > /dev/null 2>&1
python $REPO/profiles/fp64_from_pmc.py $(ls /tmp/pmc_fp64/*counter_collection.csv | head -1) $OUT/r02_kernel_stats.csv $OUT/r02_pmc_fp64_by_kernel.csv $OUT/fp64_utilisation.json
# 5. the other configurations (parity-test cases and context, not bench lines)
for c in 1 2; do timeout 600 python $REPO/bench.py --config $c --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_config${c}.json; done
timeout 900 python $REPO/bench.py --config 4 --steps 100 --warmup 20 --repeats 3 --tagging-passes 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_config4.json
timeout 600 python $REPO/bench.py --config 5 --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_config3_50hz_knots.json
timeout 900 python $REPO/bench.py --config 6 --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_euroc_shape.json
CALICO_SOLVER=band timeout 600 python $REPO/bench.py --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 > $OUT/r02_bench_band_solver.json
CALICO_SOLVER=band timeout 600 python $REPO/bench.py --config 5 --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_config3_50hz_knots_band_solver.json
timeout 300 python $REPO/bench.py --force-collective --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 > $OUT/r02_collective_1gpu.json
# 6. speculative evaluation under rejections: configs[4] (2 % gross outliers) and a poor start
for sp in 1 0; do CALICO_SPECULATIVE=$sp timeout 600 python $REPO/bench.py --config 4 --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_config4_speculative$sp.json; done
# the same A/B from a deliberately poor start (control points + 100 mrad / 100 mm of noise, focal lengths x2): ~30 % rejected steps
for sp in 1 0; do CALICO_SPECULATIVE=$sp timeout 300 python $REPO/bench.py --poor-start 100 --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_poor_start_speculative$sp.json; done
ls -la $OUT
