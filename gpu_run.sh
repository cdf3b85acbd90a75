mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_crps.py -m gpu -q -x 2>&1 | tail -12
