export TMPDIR=/tmp
mkdir -p gpurun_out/r06v
SECONDS=0
MAKANI_AMD_DIST_LOG=gpurun_out/r06v/dist.txt timeout 1800 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_dist_fullsize.py tests/test_bench_contract.py tests/test_gpu_fcn3_distributed.py -x -q -m gpu > gpurun_out/r06v/tests.log 2>&1; echo "tests rc $? at $SECONDS s"; tail -2 gpurun_out/r06v/tests.log
timeout 600 python tools/shadow_rank.py --h 4 --w 2 --steps 4 --json gpurun_out/r06v/shadow_h4w2.json > gpurun_out/r06v/shadow_h4w2.log 2>&1; python -c "
import json; d=json.load(open('gpurun_out/r06v/shadow_h4w2.json')); print({k:d[k] for k in d if 'ms' in k or 'wall' in k or 'graph' in k})"
