cd /root/repo
for pe in 1 0; do for d in 2 3; do
echo "== PREDICT_END=$pe STREAM_DEPTH=$d"
CALICO_PREDICT_END=$pe CALICO_STREAM_DEPTH=$d timeout 200 python profiles/dev/steady.py 3 20 220 5 2>&1 | tail -3
done; done
echo "== bench"
timeout 200 python bench.py --no-cpu-baseline --repeats 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['ms_per_step'])"
