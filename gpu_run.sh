mkdir -p gpurun_out/r05y
timeout 300 python tools/conv_shard_ab.py > gpurun_out/r05y/conv_shard_ab.txt 2>&1; echo rc $?
grep " px" gpurun_out/r05y/conv_shard_ab.txt | sort -k2,3 -k4n | cut -c1-160
