timeout 300 python profiles/dev/bitwise.py | tail -1
timeout 600 bash profiles/dev/ab_env.sh CALICO_FUSE_EXPAND 2
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_r; timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_r -o kt -- python /root/repo/bench.py --no-cpu-baseline --repeats 4 > /dev/null 2>&1; python /root/repo/profiles/summarize_rocpd.py /tmp/prof_r/*.db | cut -d, -f1,2,9 | cut -c1-40,100- | head -6
cd /root/repo; bash profiles/scratch_run.sh
