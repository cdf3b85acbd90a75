mkdir -p gpurun_out/r05o
timeout 200 python -m pytest tests/test_gpu_distributed.py -x -q -s -k "hipgraph or rccl" > gpurun_out/r05o/pytest_final.log 2>&1; echo "rc $?"; grep "replay \|passed\|failed" gpurun_out/r05o/pytest_final.log | cut -c1-300
