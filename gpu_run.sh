mkdir -p gpurun_out/r06p
export TMPDIR=/tmp
SECONDS=0
MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_oldfft.so python tools/fft_plan_check.py /tmp/fft_old.pt 2>&1 | grep -v amdgpu.ids | tail -1
python tools/fft_plan_check.py /tmp/fft_new.pt 2>&1 | grep -v amdgpu.ids | tail -1
python tools/fft_plan_check.py /tmp/fft_old.pt /tmp/fft_new.pt
timeout 600 python tools/ab.py run g5 g8 -- python tools/microbench.py cold pointwise > gpurun_out/r06p/gelu_slots.log 2>&1; echo "slots rc $? at $SECONDS s"; grep "bwd" gpurun_out/r06p/gelu_slots.log | cut -c1-130
timeout 300 python tools/ab.py run oldfft g5 -- python tools/microbench.py fft > gpurun_out/r06p/fft_ab.log 2>&1; grep "irfft" gpurun_out/r06p/fft_ab.log | cut -c1-130
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "instance_norm or fft" > gpurun_out/r06p/tests.log 2>&1; echo "tests rc $? at $SECONDS s"; tail -2 gpurun_out/r06p/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-sht-metric > gpurun_out/r06p/bench.json 2> gpurun_out/r06p/bench.err; echo "bench rc $? at $SECONDS s: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r06p/bench.json | head -2 | tr '\n' ' ')"
