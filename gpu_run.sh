# round-3 call 27: final bench line + rocprofv3 kernel trace + PMC passes of the default bench command
bash tools/profile_round.sh r03n > gpurun_out/r03n_profile.log 2>&1
tail -3 gpurun_out/r03n_profile.log
grep '^{' gpurun_out/r03n/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'], d['parity_rel_l2'])"
