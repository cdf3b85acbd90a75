# round-3 call 36: instance-norm apply passes with non-temporal loads, cold buffers, same box
mkdir -p gpurun_out/r03p
timeout 300 python tools/ab.py run cur pwnt -- python tools/microbench.py cold 2>&1 | grep -v "rfft\|torch" | tee gpurun_out/r03p/ab_pw_nt.txt
