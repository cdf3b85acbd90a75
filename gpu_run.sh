mkdir -p gpurun_out/final4
timeout 400 python bench.py > gpurun_out/final4/bench.json 2> gpurun_out/final4/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/final4 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric > $GRAFT_REPO_ROOT/gpurun_out/final4/bench_traced.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/final4 -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} gpurun_out/final4/kernel_stats.md > /dev/null
find gpurun_out/final4 -name "*.db" -delete
timeout 300 python tools/microbench.py > gpurun_out/final4/microbench.txt 2>&1
grep '^{' gpurun_out/final4/bench_traced.log | cut -c1-170
