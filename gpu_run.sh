# the FourCastNet3 bench line with its cpu_baseline and block-level in-run parity, on the small stand-in configuration
mkdir -p gpurun_out/r05e
SECONDS=0
timeout 100 python -m pytest tests/test_bench_contract.py -q -m gpu -s -k fcn3_workload > gpurun_out/r05e/fcn3_line.log 2>&1; echo "rc $? in $SECONDS s"
grep -v "amdgpu.ids" gpurun_out/r05e/fcn3_line.log | tail -25 | cut -c1-600
