mkdir -p gpurun_out/r02d; cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -5
for v in block panel; do echo "dense $v: $(CALICO_DENSE=$v timeout 200 python bench.py --no-cpu-baseline --repeats 7 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('it/s', d['value'])")"; done
cd /tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 20 --repeats 1 > /dev/null 2>&1
cd $R; python profiles/iteration_trace.py /tmp/kt/*.db bcr_level_kernelILb1 > gpurun_out/r02d/iter.txt; cat gpurun_out/r02d/iter.txt
