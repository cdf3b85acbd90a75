mkdir -p gpurun_out/r05r
timeout 600 python tools/glue_trace.py > gpurun_out/r05r/glue_trace.txt 2>&1; echo rc $?
grep -v "amdgpu.ids\|Warning\|warn" gpurun_out/r05r/glue_trace.txt | head -120 | cut -c1-260
