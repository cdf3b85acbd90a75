# round-3 call 34: optimizer tests on the non-temporal AdamW kernel, then the step time against the previous library
mkdir -p gpurun_out/r03p
timeout 600 python -m pytest tests/test_gpu_optim.py tests/test_gpu_kernels.py -q -x -m gpu -k "adamw or optim or zero or clip" 2>&1 | tail -3
for v in cur new cur new; do
  if [ $v = cur ]; then export MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_cur.so; else unset MAKANI_AMD_LIB; fi
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sht-metric 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['value'],3), round(d['ms_per_step'],3), d['final_loss'])"
done 2>&1 | tee gpurun_out/r03p/step_ab_adamw.txt
