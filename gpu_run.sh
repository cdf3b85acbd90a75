mkdir -p gpurun_out/r05s
timeout 600 python -m pytest tests/test_gpu_distributed.py -x -q -s -k "spatial_parallel_sfno" > gpurun_out/r05s/toy.log 2>&1; echo rc $?
grep "rank \|passed\|failed\|Error" gpurun_out/r05s/toy.log | cut -c1-200 | head -40
