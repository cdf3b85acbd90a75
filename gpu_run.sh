mkdir -p gpurun_out/prof1
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof1 -o r01 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sht-metric > gpurun_out/prof1/bench.log 2>&1
echo "prof rc=$?"
ls -R gpurun_out/prof1 | head -30
