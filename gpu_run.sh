# round-3 call 15: the round's SFNO profile set on the final code (after the FFT addressing change)
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/profile_round.sh r03j > gpurun_out/r03j_profile.log 2>&1
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03j/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "peak_hbm_GB")}, {k: v for k, v in d["parity_rel_l2"].items() if k != "what"})
print({k: d["roofline"][k] for k in ("kernel", "frac", "achieved", "ms_avg", "traffic")})
print(d["cpu_baseline"]["sample"]); print(d["fwd_sht"])
PY
python tools/segfft_bench.py | tail -3
