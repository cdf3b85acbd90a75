# round-3 call 42: FourCastNet3 bench line on the final library
mkdir -p gpurun_out/r03s
timeout 400 python bench.py --config fcn3_sc2_edim45_layers10 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03s/bench_fcn3.json 2> gpurun_out/r03s/bench_fcn3.err
grep '^{' gpurun_out/r03s/bench_fcn3.json | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('peak_hbm_GB'))"
