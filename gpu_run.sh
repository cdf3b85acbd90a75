# round-3 call 22: dead-row waves skipped (ds) against the same sources without it (pk4) and the earlier build (pk2)
mkdir -p gpurun_out/r03l
timeout 600 python tools/ab.py run pk2 pk4 ds -- python tools/microbench.py dhconv 2>&1 | grep -v gen1 | tee gpurun_out/r03l/ab_deadrows.txt
