# round-3 call 2: the new parity tests (all of them), CRPS cdf, FCN3 bench workload
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_headline.py tests/test_gpu_disco.py tests/test_fcn3.py tests/test_gpu_optim.py tests/test_gpu_distributed.py tests/test_crps.py tests/test_bench_contract.py -m gpu -q -s \
  -k "config2 or block0 or block7 or fcn3_grids or decoder_grid or local_block_360 or zero1 or resumes or ragged or crps or fcn3_workload" 2>&1 | grep -v "^$" | tail -120 > gpurun_out/r03b_newtests.log
tail -70 gpurun_out/r03b_newtests.log
timeout 900 python bench.py --config fcn3_sc2_edim45_layers10 --steps 5 --warmup 2 > gpurun_out/r03b_bench_fcn3.json 2> gpurun_out/r03b_bench_fcn3.err
tail -5 gpurun_out/r03b_bench_fcn3.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r03b_bench_fcn3.json") if l.startswith("{")][-1])
    print({k: d.get(k) for k in ("metric", "value", "ms_per_step", "peak_hbm_GB", "final_loss", "note")})
    print(d["roofline"])
    print({k: v["ms_per_step"] for k, v in list(d["hip_kernels"].items())[:25]})
except Exception as e:
    print("no bench line", e)
PY
