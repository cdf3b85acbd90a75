# round-5 call 10: with disjoint compute units per rank: distributed-vs-serial diagnostic, shadow rank (merged plane blocks), full-size tests
mkdir -p gpurun_out/r05j
export MAKANI_AMD_DIST_LOG=$PWD/gpurun_out/r05j/dist_fullsize.txt
timeout 300 python tools/shadow_rank.py --h 4 --w 2 --steps 3 --json gpurun_out/r05j/shadow_h4w2.json > gpurun_out/r05j/shadow_h4w2.log 2>&1; echo "shadow rc $?"
tail -1 gpurun_out/r05j/shadow_h4w2.log | cut -c1-900
timeout 300 python tools/dist_diag.py --h 4 --w 1 --variants base > gpurun_out/r05j/diag_h4w1.log 2>&1; echo "diag h4w1 rc $?"
grep "^\[base\]\|encoder.fwd.2\|blocks.0.filter\|blocks.7.filter\|outer_skip" gpurun_out/r05j/diag_h4w1.log | head -30
timeout 300 python tools/dist_diag.py --h 4 --w 2 --variants base > gpurun_out/r05j/diag_h4w2.log 2>&1; echo "diag h4w2 rc $?"
grep "^\[base\]\|encoder.fwd.2\|blocks.0.filter\|blocks.7.filter" gpurun_out/r05j/diag_h4w2.log | head -30
timeout 1500 python -m pytest tests/test_gpu_dist_fullsize.py -x -q -s --durations=10 > gpurun_out/r05j/pytest.log 2>&1; echo "pytest rc $?"
tail -15 gpurun_out/r05j/pytest.log
cat gpurun_out/r05j/dist_fullsize.txt | cut -c1-400
