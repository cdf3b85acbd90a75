# round-3 call 41: full GPU suite + smoke on the final library
mkdir -p gpurun_out/r03r
timeout 1500 python -m pytest tests -q -x -m gpu --durations=5 2>&1 | tail -12 | tee gpurun_out/r03r/gpu_suite_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee -a gpurun_out/r03r/gpu_suite_tail.txt
