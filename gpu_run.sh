# round-3 call 1: the new parity tests + the bench line with parity_rel_l2
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_headline.py tests/test_gpu_disco.py tests/test_fcn3.py tests/test_gpu_optim.py tests/test_gpu_distributed.py -m gpu -q -x -s \
  -k "config2 or block0 or block7 or fcn3_grids or decoder_grid or local_block_360 or zero1 or resumes or ragged" 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r03a_newtests.log
tail -30 gpurun_out/r03a_newtests.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
tail -3 gpurun_out/r03a_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03a_bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "parity_rel_l2", "peak_hbm_GB")})
print(d["cpu_baseline"])
PY
