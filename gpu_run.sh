mkdir -p gpurun_out/r06r
export TMPDIR=/tmp
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06r/bench_driver_form.json 2> gpurun_out/r06r/bench_driver_form.err; echo "driver-form bench rc $? took $SECONDS s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06r/bench_driver_form.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["exact_fp32_spectral"]["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic_source"][:60], d["cpu_baseline"]["value"])
PY
