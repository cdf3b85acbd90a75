#!/bin/bash
# round-4 call 6: where the weight-stationary GEMM's cycles go (s_memtime stamps); the default bench line with the gradient parity keys
O=gpurun_out/r04f; mkdir -p $O
MAKANI_AMD_LIB=$PWD/makani_amd/libmakani_amd_adiag.so timeout 600 python tools/astat_diag.py > $O/astat_diag.txt 2>&1; cat $O/astat_diag.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['fwd_sht'])
print(json.dumps({k:v for k,v in d['parity_rel_l2'].items() if k!='what'}, indent=0))"
