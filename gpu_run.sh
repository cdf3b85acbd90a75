mkdir -p gpurun_out/prof3
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof3 -o r01c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sht-metric > gpurun_out/prof3/bench.log 2>&1
echo "prof rc=$?"; grep '^{' gpurun_out/prof3/bench.log | cut -c1-200
