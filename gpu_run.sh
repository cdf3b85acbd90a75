# the last GPU call of round 5: the whole GPU suite on the final tree, then smoke (tools/r05_final.sh lists how every record was made)
mkdir -p gpurun_out/r05u
export MAKANI_AMD_DIST_LOG=$PWD/gpurun_out/r05u/dist_fullsize.txt
SECONDS=0
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r05u/gpu_suite.log 2>&1; echo "pytest rc $? in $SECONDS s"
tail -22 gpurun_out/r05u/gpu_suite.log | cut -c1-200
python __graft_entry__.py smoke 2>&1 | tail -2
# configs[4] per rank: the 4-step rollout on one rank of h4 w2 (phantom collectives) and serially
for cfg in "1 1" "4 2"; do set -- $cfg
  timeout 400 python tools/shadow_rank.py --h $1 --w $2 --steps 3 --multistep-count 4 --json gpurun_out/r05u/shadow_ms4_h$1w$2.json > gpurun_out/r05u/shadow_ms4_h$1w$2.log 2>&1; echo "shadow ms4 h$1w$2 rc $?"
  tail -1 gpurun_out/r05u/shadow_ms4_h$1w$2.log | cut -c1-700
done
