export TMPDIR=/tmp
mkdir -p gpurun_out/r06t
timeout 600 python tools/ab.py run occ0 occ1 -- python tools/microbench.py fft cold > gpurun_out/r06t/fft_occ.log 2>&1; grep "irfft" gpurun_out/r06t/fft_occ.log | cut -c1-140
for t in occ0 occ1; do MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_$t.so python tools/fft_plan_check.py /tmp/fft_$t.pt 2>&1 | grep -v amdgpu | tail -1; done; python tools/fft_plan_check.py /tmp/fft_occ0.pt /tmp/fft_occ1.pt
