mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/segfft_bench.py 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "fft or sht or segmented" 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
for k in ('rfft_1440','irfft_1440','rfft_480','irfft_480'):
    print(k, d['hip_kernels'][k])
"
