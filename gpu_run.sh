# last GPU call of round 5: the distributed / full-size / bench-contract GPU tests on the final tree (the whole suite ran two commits
# earlier: profiles/r05_gpu_suite.txt; the kernel-side commits since then were covered by the 59 channel-GEMM / FCN3 tests of call 30)
mkdir -p gpurun_out/r05zz
export MAKANI_AMD_DIST_LOG=$PWD/gpurun_out/r05zz/dist_fullsize.txt
SECONDS=0
timeout 700 python -m pytest tests/test_gpu_dist_fullsize.py tests/test_gpu_distributed.py tests/test_gpu_optim.py tests/test_bench_contract.py -x -q -m gpu > gpurun_out/r05zz/pytest.log 2>&1; echo "pytest rc $? in $SECONDS s"
tail -3 gpurun_out/r05zz/pytest.log
python __graft_entry__.py smoke 2>&1 | tail -1
