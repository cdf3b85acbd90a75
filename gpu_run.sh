mkdir -p gpurun_out/r06q
export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "side_stream or rollout or tf32" > gpurun_out/r06q/tests.log 2>&1; echo "tests rc $? at $SECONDS s"; tail -5 gpurun_out/r06q/tests.log | cut -c1-300
for ov in "" "--no-wgrad-overlap"; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-sht-metric --no-exact $ov > gpurun_out/r06q/bench$ov.json 2> gpurun_out/r06q/bench$ov.err; echo "bench [$ov] rc $? at $SECONDS s: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r06q/bench$ov.json | head -1) $(grep -o '"launch": "[a-zA-Z ]*' gpurun_out/r06q/bench$ov.json)"; done
grep "graph capture failed" gpurun_out/r06q/*.err | cut -c1-300
