# round-3 call 9: tests touched since call 3 + the round's SFNO profile set on the current code
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_crps.py tests/test_bench_contract.py tests/test_gpu_kernels.py tests/test_gpu_optim.py -m gpu -q -x \
   -k "crps or bench or shadow or segmented or zero1 or resumes or layernorm or fp32_channel" > gpurun_out/r03h_tests.log 2>&1
tail -6 gpurun_out/r03h_tests.log
bash tools/profile_round.sh r03h > gpurun_out/r03h_profile.log 2>&1
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03h/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "peak_hbm_GB")}, {k: v for k, v in d["parity_rel_l2"].items() if k != "what"})
print({k: d["roofline"][k] for k in ("kernel", "frac", "achieved", "ms_avg", "traffic")})
print(d["cpu_baseline"])
PY
tail -5 gpurun_out/r03h/bench.err
