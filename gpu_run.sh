# round-3 call 3: full GPU suite on the reworked fp32 path + FCN3 bench with the 720-point fast FFT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s -x --durations=15 > gpurun_out/r03c_gpu_suite.log 2>&1
grep -n "rel-L2\|FCN3 local\|passed\|failed\|^FAILED\|^ERROR\|Error\|slowest\|s call\|s setup" gpurun_out/r03c_gpu_suite.log | tail -80
timeout 900 python bench.py --config fcn3_sc2_edim45_layers10 --steps 5 --warmup 2 > gpurun_out/r03c_bench_fcn3.json 2> gpurun_out/r03c_bench_fcn3.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r03c_bench_fcn3.json") if l.startswith("{")][-1])
    print({k: d.get(k) for k in ("metric", "value", "ms_per_step", "peak_hbm_GB", "final_loss", "note")})
    print({k: v["ms_per_step"] for k, v in list(d["hip_kernels"].items())[:14]})
except Exception as e:
    print("no bench line", e)
PY
