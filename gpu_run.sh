# round-3 call 28: full GPU suite + smoke on the final library (two-pass 1440-point FFT, trimmed headers)
mkdir -p gpurun_out/r03o
timeout 1500 python -m pytest tests -q -x -m gpu --durations=5 2>&1 | tail -12 | tee gpurun_out/r03o/gpu_suite_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee -a gpurun_out/r03o/gpu_suite_tail.txt
