# round-3 call 37: LDS-DMA streams of the channel GEMMs with the non-temporal policy: conv tests, then step time same box
mkdir -p gpurun_out/r03p
MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_dnt.so timeout 600 python -m pytest tests/test_gpu_headline.py tests/test_gpu_kernels.py -q -x -m gpu -k "conv1x1 or instnorm or instance" 2>&1 | tail -2
for v in cur dnt cur dnt; do
  export MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_$v.so
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sht-metric 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['value'],3), round(d['ms_per_step'],3), d['final_loss'])"
done 2>&1 | tee gpurun_out/r03p/step_ab_dnt.txt
