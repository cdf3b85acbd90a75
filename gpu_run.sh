# the FourCastNet3 line's cpu_baseline on the GPU box's host cores (no GPU work: the child process of bench.py alone)
mkdir -p gpurun_out/r05c
SECONDS=0
nproc > gpurun_out/r05c/host.txt; free -g | head -2 >> gpurun_out/r05c/host.txt
timeout 240 python -c "
import json, sys
sys.argv=['bench.py']
import bench
print(json.dumps(bench.cpu_baseline('fcn3_sc2_edim45_layers10', timeout_s=200)))
" > gpurun_out/r05c/fcn3_cpu_baseline.json 2> gpurun_out/r05c/err.log
echo "rc $? in $SECONDS s"; cat gpurun_out/r05c/host.txt; cut -c1-1500 gpurun_out/r05c/fcn3_cpu_baseline.json
