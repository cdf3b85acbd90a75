# round-3 call 35: pre-activation with non-temporal stores (weight-stationary channel GEMM), step time same box
mkdir -p gpurun_out/r03p
for v in cur pnt cur pnt; do
  export MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_$v.so
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sht-metric 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['value'],3), round(d['ms_per_step'],3), d['final_loss'])"
done 2>&1 | tee gpurun_out/r03p/step_ab_pnt.txt
