mkdir -p gpurun_out/r06h
export TMPDIR=/tmp
timeout 300 python tools/pk_hazard_probe.py --reps 3 --culprit chan_gemm_f32 > gpurun_out/r06h/pk_forms.log 2>&1; echo "probe rc $?"
grep -v "amdgpu.ids" gpurun_out/r06h/pk_forms.log | grep "chan_gemm" | grep -v "gap s_nop" | tail -6 | cut -c1-160
