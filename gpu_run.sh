# round 6, call 12: per-kind slots of the one-pass norm (tests, bench A/B), FourCastNet3 h2w2 at full size, contractive bf16 rollout
mkdir -p gpurun_out/r06l
export TMPDIR=/tmp
SECONDS=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "instance_norm" > gpurun_out/r06l/norm_tests.log 2>&1; echo "norm tests rc $? at $SECONDS s"; tail -3 gpurun_out/r06l/norm_tests.log
for f in 1 0 1 0; do MAKANI_AMD_NORM_FUSED=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-sht-metric > gpurun_out/r06l/bench_fused$f.log 2>&1; echo "bench fused=$f rc $? at $SECONDS s: $(grep -o '"ms_per_step": [0-9.]*, "higher' gpurun_out/r06l/bench_fused$f.log | head -1)"; done
MAKANI_AMD_DIST_LOG=gpurun_out/r06l/fcn3_fullsize.txt timeout 2400 python -m pytest tests/test_gpu_fcn3_fullsize.py -x -q -m gpu > gpurun_out/r06l/fcn3_full.log 2>&1; echo "fcn3 full-size h2w2 rc $? at $SECONDS s"; tail -5 gpurun_out/r06l/fcn3_full.log | cut -c1-300; cut -c1-330 gpurun_out/r06l/fcn3_fullsize.txt
MAKANI_AMD_DIST_LOG=gpurun_out/r06l/dist_fullsize.txt timeout 1800 python -m pytest tests/test_gpu_dist_fullsize.py -x -q -m gpu -k "multistep4" > gpurun_out/r06l/rollout.log 2>&1; echo "rollout rc $? at $SECONDS s"; tail -5 gpurun_out/r06l/rollout.log | cut -c1-300; grep "^---\|rank 0:" gpurun_out/r06l/dist_fullsize.txt | cut -c1-380
