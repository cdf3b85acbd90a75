# the reference's own model properties on the HIP path (tests/test_models.py: shapes, gradient accumulation)
mkdir -p gpurun_out/r05d
SECONDS=0
timeout 100 python -m pytest tests/test_gpu_reference_properties.py -q -m gpu -s > gpurun_out/r05d/props.log 2>&1; echo "rc $? in $SECONDS s"
grep -v "amdgpu.ids" gpurun_out/r05d/props.log | tail -25
