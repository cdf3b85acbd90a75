mkdir -p gpurun_out/pmc7
R=$GRAFT_REPO_ROOT
timeout 300 python tools/microbench.py > gpurun_out/pmc7/microbench.log 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc7 -o $c -- python $R/tools/microbench.py fft legendre dhconv conv > $R/gpurun_out/pmc7/$c.log 2>&1
  echo "$c rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc7/summary.md $(find gpurun_out/pmc7 -name "*counter_collection.csv") > /dev/null
find gpurun_out/pmc7 -name "*.csv" -delete
grep -v amdgpu gpurun_out/pmc7/microbench.log | cut -c1-170
