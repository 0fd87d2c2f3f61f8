mkdir -p gpurun_out/r02d; cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -x -q -m gpu 2>&1 | tail -5
CALICO_KERNEL_TIMING=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 10 2>&1 | grep "^bcr_back" | tail -2
for q in 1 2 3 4 5 6; do echo "leaf $q: $(CALICO_BCR_LEAF=$q timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('it/s', d['value'])")"; done
cd /tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 20 > /dev/null 2>&1
cd $R; python profiles/iteration_trace.py /tmp/kt/*.db bcr_level_kernelILb1 > gpurun_out/r02d/iter.txt; cat gpurun_out/r02d/iter.txt
