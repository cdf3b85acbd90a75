# round-3 call 45: final library: the 384-channel block tests (240x480, bf16 + fp32) and the ragged distributed split
mkdir -p gpurun_out/r03s
timeout 400 python -m pytest tests/test_gpu_headline.py tests/test_gpu_distributed.py -q -x -m gpu -k "block_240x480 or ragged" 2>&1 | tail -3 | tee gpurun_out/r03s/final_subset2.txt
