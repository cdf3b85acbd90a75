mkdir -p gpurun_out/final
timeout 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -c 600 gpurun_out/final/bench.json | head -c 400; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/final -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric > $GRAFT_REPO_ROOT/gpurun_out/final/bench_traced.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/final -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} gpurun_out/final/kernel_stats.md > /dev/null
find gpurun_out/final -name "*.db" -delete
timeout 300 python tools/microbench.py > gpurun_out/final/microbench.txt 2>&1
head -5 gpurun_out/final/kernel_stats.md | cut -c1-120
