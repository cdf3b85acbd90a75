# round-3 call 38: dhconv weight gradient with non-temporal result stores, same box
mkdir -p gpurun_out/r03p
timeout 300 python tools/ab.py run cur wnt cur wnt -- python tools/microbench.py dhconv 2>&1 | grep "gen2 dhconv wgrad" | tee gpurun_out/r03p/ab_wgrad_nt.txt
