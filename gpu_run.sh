# round-3 call 48: the default bench command on the final library (without the CPU baseline child)
mkdir -p gpurun_out/r03s
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/r03s/bench_final.err | grep '^{' | tail -1 > gpurun_out/r03s/bench_final.json
python -c "import json; d=json.load(open('gpurun_out/r03s/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['final_loss'], d['fwd_sht'])"
