# round-3 call 40: final bench line + rocprofv3 kernel trace + PMC passes of the default bench command
bash tools/profile_round.sh r03q > gpurun_out/r03q_profile.log 2>&1
tail -2 gpurun_out/r03q_profile.log
grep '^{' gpurun_out/r03q/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['ms_per_step'], d['parity_rel_l2']['fp32'], d['parity_rel_l2']['bf16_autocast'])"
