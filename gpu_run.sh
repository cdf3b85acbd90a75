# FourCastNet3 under bf16 autocast with the other filter bases (5 / 6 basis functions), forward + backward
mkdir -p gpurun_out/r05h
SECONDS=0
timeout 60 python -m pytest tests/test_fcn3.py -q -m gpu -s -k "bf16" > gpurun_out/r05h/bf16.log 2>&1; echo "rc $? in $SECONDS s"
grep -v "amdgpu.ids" gpurun_out/r05h/bf16.log | tail -30 | cut -c1-300
