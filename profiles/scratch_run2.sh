timeout 300 python profiles/dev/bitwise.py | tail -1
timeout 600 bash profiles/dev/ab_env.sh CALICO_FUSE_EXPAND 2
bash profiles/scratch_run.sh
