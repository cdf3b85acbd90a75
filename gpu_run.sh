mkdir -p gpurun_out/prof7
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric 2>&1 | tail -1 | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof7 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 0 --no-cpu-baseline --no-sht-metric > $GRAFT_REPO_ROOT/gpurun_out/prof7/bench.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof7 -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} gpurun_out/prof7/kernel_stats.md > /dev/null
find gpurun_out/prof7 -name "*.db" -delete
tail -1 gpurun_out/prof7/bench.log | cut -c1-200
