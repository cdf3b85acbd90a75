#!/bin/bash
O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1x1 or conv_gelu or weight_stationary or instance_norm or rfft or irfft or layernorm or bias_gelu" > $O/kernels.log 2>&1; tail -3 $O/kernels.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sht-metric"
step() { "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['hip_kernels']; f=lambda p: round(sum(v['ms_per_step'] for n,v in k.items() if n.startswith(p)),3); print(round(d['ms_per_step'],3), 'ms/step; nn', f('conv1x1_nn'), 'instnorm', f('instnorm'), 'irfft', f('irfft'), 'loss', d['final_loss'])"; }
{ for i in 1 2; do echo "== v_cvt_pk_bf16_f32 pair packing"; step $B; echo "== before (libmakani_amd_prev.so)"; MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_prev.so step $B; done; } > $O/step_ab_pack.txt 2>&1; cat $O/step_ab_pack.txt
