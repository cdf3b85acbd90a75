# last check of the round: the driver's bench command on the final tree (short form: no CPU child, no counter passes)
mkdir -p gpurun_out/r05f
SECONDS=0
timeout 70 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > gpurun_out/r05f/bench_short.json 2> gpurun_out/r05f/err.log; echo "rc $? in $SECONDS s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05f/bench_short.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('metric','value','ms_per_step','steps','warmup','dtype')}, d['roofline']['kernel'], d['roofline']['frac'], d['config']['workload'])
PY
