# round-5 call 20: the whole GPU suite (as the driver runs it) with durations, then smoke
mkdir -p gpurun_out/r05p
export MAKANI_AMD_DIST_LOG=$PWD/gpurun_out/r05p/dist_fullsize.txt
SECONDS=0
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=40 > gpurun_out/r05p/gpu_suite.log 2>&1; echo "pytest rc $? in $SECONDS s"
tail -60 gpurun_out/r05p/gpu_suite.log | cut -c1-200
python __graft_entry__.py smoke 2>&1 | tail -2
