mkdir -p gpurun_out/r02d; cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED" | tail -3
for v in 1 0; do echo "fuse_top $v: $(CALICO_FUSE_TOP=$v timeout 200 python bench.py --no-cpu-baseline --repeats 9 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")"; done
cd /tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 20 --repeats 1 > /dev/null 2>&1
cd $R; python profiles/iteration_trace.py /tmp/kt/*.db bcr_level_kernelILb1 > gpurun_out/r02d/iter.txt; cat gpurun_out/r02d/iter.txt
