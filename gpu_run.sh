# round-3 call 11: the full GPU suite + smoke on the final code
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r03i_gpu_suite.log 2>&1
tail -14 gpurun_out/r03i_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
