mkdir -p gpurun_out/r06n
export TMPDIR=/tmp
SECONDS=0
MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_oldfft.so python tools/fft_plan_check.py /tmp/fft_old.pt 2>&1 | grep -v amdgpu.ids | tail -1
MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_newfft.so python tools/fft_plan_check.py /tmp/fft_new.pt 2>&1 | grep -v amdgpu.ids | tail -1
python tools/fft_plan_check.py /tmp/fft_old.pt /tmp/fft_new.pt
timeout 600 python tools/ab.py run oldfft newfft -- python tools/microbench.py fft cold > gpurun_out/r06n/fft_ab.log 2>&1; echo "fft ab rc $? at $SECONDS s"; grep "fft" gpurun_out/r06n/fft_ab.log | cut -c1-150
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/r06n/bench.json 2> gpurun_out/r06n/bench.err; echo "bench rc $? at $SECONDS s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06n/bench.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "launch", d["config"]["launch"], "spectral", d["config"]["spectral_arithmetic"])
print("exact", d.get("exact_fp32_spectral"))
print("roofline", {k: d["roofline"][k] for k in ("kernel","achieved","frac","bound")} )
print("fwd_sht", d.get("fwd_sht"))
for k,v in sorted(d.get("hip_kernels",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:16]: print("   %-44s %7.3f ms/step  %s" % (k[:44], v["ms_per_step"], v.get("launches_per_step")))
PY
MAKANI_AMD_DIST_LOG=gpurun_out/r06n/dist_fullsize.txt timeout 1800 python -m pytest tests/test_gpu_dist_fullsize.py -x -q -m gpu -k "multistep4" > gpurun_out/r06n/rollout.log 2>&1; echo "rollout rc $? at $SECONDS s"; tail -4 gpurun_out/r06n/rollout.log | cut -c1-300; grep "^---\|rank 0:" gpurun_out/r06n/dist_fullsize.txt | cut -c1-400
