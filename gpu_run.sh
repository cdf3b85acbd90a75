# round-5 call 1: full-size distributed tests (configs 3 / 5 on N ranks sharing the GPU) + shard-shape tables (shadow ranks)
mkdir -p gpurun_out/r05a
export MAKANI_AMD_DIST_LOG=$PWD/gpurun_out/r05a/dist_fullsize.txt
date +%T > gpurun_out/r05a/times.txt
( timeout 1300 python -m pytest tests/test_gpu_dist_fullsize.py -x -q -s --durations=10 > gpurun_out/r05a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05a/times.txt; date +%T >> gpurun_out/r05a/times.txt ) &
PT=$!
for cfg in "1 1" "4 2" "4 1" "2 1"; do
  set -- $cfg
  timeout 300 python tools/shadow_rank.py --h $1 --w $2 --steps 3 --json gpurun_out/r05a/shadow_h$1w$2.json > gpurun_out/r05a/shadow_h$1w$2.log 2>&1
  echo "shadow h$1w$2 rc $? $(date +%T)" >> gpurun_out/r05a/times.txt
done
wait $PT
tail -5 gpurun_out/r05a/pytest.log
cat gpurun_out/r05a/times.txt
