#!/bin/bash
# scratch: store cache policy A/B per kernel family (kernel-trace sums per family + plain bench)
O=$PWD/gpurun_out/r07n; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for t in b0 pw ff x2; do
  MAKANI_AMD_LIB=$R/makani_amd/libmakani_amd_$t.so timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sht-metric --no-pmc --no-exact > $O/kt_$t.log 2>&1
  find $O/kt_$t -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$t.csv \;
  rm -rf $O/kt_$t
done
cd $R
for t in b0 pw ff x2 b0 pw ff x2; do
  MAKANI_AMD_LIB=$R/makani_amd/libmakani_amd_$t.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-exact --no-sht-metric 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', round(d['ms_per_step'],3), d['final_loss'])" >> $O/bench.txt 2>&1; done
python - <<'PY'
import csv,re
fam=[("conv fwd/dgrad", r"conv_nn_"), ("conv wgrad", r"conv_wgrad_|reduce_splits"), ("dhconv", r"xcgemm2?_kernel"), ("Legendre", r"xgemm2?_kernel"), ("FFT", r"fft_(fast_)?kernel"), ("norm", r"in_(stats|apply|bwd|fwd)"), ("AdamW+clip", r"adamw|sumsq|clip_coef|gather_part"), ("glue", r"at::native|rocclr")]
print("variant  total  "+"  ".join(f[0] for f in fam))
for t in ("b0","pw","ff","x2"):
    rows=list(csv.DictReader(open(f'gpurun_out/r07n/kernel_stats_{t}.csv')))
    tot=sum(float(r['TotalDurationNs']) for r in rows)/18e6
    out=[]
    for name,rx in fam:
        out.append(sum(float(r['TotalDurationNs']) for r in rows if re.search(rx,r['Name']))/18e6)
    print(t, f"{tot:7.3f}", "  ".join(f"{v:7.3f}" for v in out))
PY
cat $O/bench.txt
