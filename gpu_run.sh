mkdir -p gpurun_out
SECONDS=0
timeout 280 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
echo "rc=$? elapsed=${SECONDS}s"; grep "^\[bench\]" gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step']); print(d['roofline']); print(d['fwd_sht']); print(d['cpu_baseline'])"
