# round-3 call 46: CRPS kernels with five instead of thirteen compiled ensemble capacities
mkdir -p gpurun_out/r03s
timeout 400 python -m pytest tests/test_crps.py tests/test_losses.py tests/test_fcn3.py -q -x -m gpu 2>&1 | tail -3 | tee gpurun_out/r03s/crps_tests.txt
