# round-5 final measurement: the default bench line (live counter passes, CPU baseline with bf16 yardsticks), rocprofv3 kernel trace,
# SQ counters, the FourCastNet3 line, and the bench-contract GPU tests on the final bench.py
bash tools/profile_round.sh r05z fcn3 > gpurun_out/r05z_profile.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05z/bench.json'))
r=d['roofline']
print('SFNO', round(d['value'],3), round(d['ms_per_step'],3), r['kernel'], r['frac'], r['traffic'], str(r.get('traffic_source'))[:40], d['cpu_baseline']['value'], d['fwd_sht'])
print({k: (round(v,6) if isinstance(v,float) else v) for k,v in d['parity_rel_l2'].items() if k not in ('what','bf16_gate')})
try:
    f=json.load(open('gpurun_out/r05z/bench_fcn3.json'))
    r=f['roofline']; print('FCN3', round(f['value'],3), round(f['ms_per_step'],2), r['kernel'], r['frac'], r['traffic'], str(r.get('traffic_source'))[:40])
except Exception as e:
    print('fcn3 line missing', e)
PY
timeout 600 python -m pytest tests/test_bench_contract.py -x -q -m gpu > gpurun_out/r05z/pytest_bench_contract.log 2>&1; echo "bench contract rc $?"; tail -2 gpurun_out/r05z/pytest_bench_contract.log
